"""Host-side mirror of the reference's descriptor matchers (k = 2 Hamming).

  BinaryDescriptorMatcher.knnMatch  <->  cv::line_descriptor_c::BinaryDescriptorMatcher::knnMatch
      Thirdparty/line_descriptor/src/binary_descriptor_matcher_custom.cpp:258-336
      (called from LineMatcher::ComputeDescriptorMatches, src/LineMatcher.cc:2568-2622)
  BFMatcherHamming.knnMatch         <->  cv::BFMatcher(NORM_HAMMING).knnMatch (src/Frame.cc:2977)

Same argument meaning and error behaviour as the reference: empty inputs or a
mask of the wrong shape print a message and return without touching `matches`.
"""
from collections import namedtuple

import numpy as np
import torch

from . import _lib

DMatch = namedtuple("DMatch", ["queryIdx", "trainIdx", "imgIdx", "distance"])


def knn2_raw(query, train, qmask=None, tie_rule=_lib.TIE_MIH):
    """(idx[nq,2], dist[nq,2]) int32.  numpy in -> numpy out (host flavour of the
    C ABI), torch cuda tensors in -> torch cuda tensors out (device flavour,
    asynchronous on the current stream)."""
    if isinstance(query, torch.Tensor):
        assert query.is_cuda and train.is_cuda and query.dtype == torch.uint8
        nq, nt = query.shape[0], train.shape[0]
        idx = torch.empty((nq, 2), dtype=torch.int32, device=query.device)
        dist = torch.empty((nq, 2), dtype=torch.int32, device=query.device)
        _lib.check(_lib.lib.plvs_hip_hamming_knn2_dev(
            _lib.t_ptr(query), nq, _lib.t_ptr(train), nt, _lib.t_ptr(qmask), tie_rule,
            _lib.t_ptr(idx), _lib.t_ptr(dist), _lib.current_stream_ptr()))
        return idx, dist
    query = np.ascontiguousarray(query, dtype=np.uint8)
    train = np.ascontiguousarray(train, dtype=np.uint8)
    nq, nt = query.shape[0], train.shape[0]
    idx = np.empty((nq, 2), dtype=np.int32)
    dist = np.empty((nq, 2), dtype=np.int32)
    if qmask is not None:
        qmask = np.ascontiguousarray(qmask, dtype=np.uint8).reshape(-1)
    _lib.check(_lib.lib.plvs_hip_hamming_knn2(
        _lib.np_ptr(query), nq, _lib.np_ptr(train), nt, _lib.np_ptr(qmask), tie_rule,
        _lib.np_ptr(idx), _lib.np_ptr(dist)))
    return idx, dist


class _KnnMatcher:
    tie_rule = _lib.TIE_LOWEST_INDEX

    def knnMatch(self, queryDescriptors, trainDescriptors, matches, k=2, mask=None,
                 compactResult=False):
        """Appends one list of k DMatch per query to `matches` (list of lists)."""
        q = np.asarray(queryDescriptors)
        t = np.asarray(trainDescriptors)
        if q.shape[0] == 0 or t.shape[0] == 0:
            print("Error: descriptors matrices cannot be void")
            return
        if mask is not None:
            mask = np.asarray(mask)
            if mask.size and (mask.shape[0] != q.shape[0] or (mask.ndim > 1 and mask.shape[1] != 1)):
                print(f"Error: input mask should have {q.shape[0]} rows and 1 column. "
                      "Program will be terminated")
                return
            if mask.size == 0:
                mask = None
        if k != 2:
            raise NotImplementedError("the accelerated path implements k = 2 (the only value PLVS uses)")
        idx, dist = knn2_raw(q, t, mask, self.tie_rule)
        for i in range(q.shape[0]):
            if mask is not None and mask.reshape(-1)[i] == 0:
                if not compactResult:
                    matches.append([])
                continue
            matches.append([DMatch(i, int(idx[i, j]), 0, float(dist[i, j])) for j in range(2)
                            if idx[i, j] >= 0])


class BinaryDescriptorMatcher(_KnnMatcher):
    """LBD matcher; ties follow the reference's multi-index-hash discovery order."""
    tie_rule = _lib.TIE_MIH


class BFMatcherHamming(_KnnMatcher):
    """cv::BFMatcher(NORM_HAMMING): ties go to the lowest train index."""
    tie_rule = _lib.TIE_LOWEST_INDEX
