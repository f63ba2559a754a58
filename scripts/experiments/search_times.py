#!/usr/bin/env python3
"""Median time per call of the six search functions (host flavours) at bench.py's sizes, beside the CPU restatement."""
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from plvs_amd.linematcher import LineMatcher, line_frame_view  # noqa: E402
from plvs_amd.orbmatcher import ORBmatcher  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests import test_line_proj_search as tlp  # noqa: E402
from tests import test_line_search as tls  # noqa: E402
from tests import test_orb_search as tos  # noqa: E402

ora = oracle_lib.load()


def us(fn, reps=40):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e6, 1), round(float(np.min(ts)) * 1e6, 1)


F, M, occ = tos.make_case(1, n=2000, m=1500)
om = ORBmatcher(0.8, True)
print("orb mappoints", us(lambda: om.SearchByProjection(F, M, 1.0, False, 40.0, occupied=occ)),
      "cpu", us(lambda: tos.oracle_search(ora.lib, F, M, 1.0, False, 40.0, 0.8, occ), 10), flush=True)
F2, ang, mx, my, mbf, L, occ2 = tos.make_ff_case(1, n=2000)
om2 = ORBmatcher(0.9, True)
print("orb lastframe", us(lambda: om2.SearchByProjectionLastFrame(F2, ang, mx, my, mbf, L, 15.0, False, False, occupied=occ2)),
      "cpu", us(lambda: tos.oracle_search_ff(ora.lib, F2, ang, mx, my, mbf, L, 15.0, 0, 0, 1, occ2), 10), flush=True)
KV, kd, kv, ka, FV, fd, fa = tos.make_bow_case(1, nk=2000, nf=2000)
om3 = ORBmatcher(0.7, True)
print("orb bow", us(lambda: om3.SearchByBoW(KV, kd, kv, ka, FV, fd, fa)),
      "cpu", us(lambda: tos.oracle_search_bow(ora.lib, KV, kd, kv, ka, FV, fd, fa, 0.7, 1), 10), flush=True)
lc = tls.make_case(1, n_last=250, n_cur=300)
lm = LineMatcher(0.8, True)
print("lines knn kf", us(lambda: lm.SearchByKnn(lc[0], lc[1], lc[2], lc[3], lc[4])),
      "cpu", us(lambda: tls.run(tls.oracle_fn(ora), lc, 0.8, True), 5), flush=True)
pc = tlp.make_case(1, n_cur=300, n_last=250)
view = line_frame_view(pc["kl"], pc["desc"], tlp.SCALE, tlp.INV_SIGMA2, tlp.MAX_DIAG)
print("lines proj lastframe", us(lambda: lm.SearchByProjectionLastFrame(view, pc["valid"], pc["proj"], pc["octave"], pc["angle"], pc["ldesc"],
                                                                        occupied=pc["occupied"], has_obs=pc["has_obs"])),
      "cpu", us(lambda: tlp.oracle_ff(ora, pc, False, 0, 0.8, True), 10), flush=True)
print("lines proj maplines", us(lambda: lm.SearchByProjection(view, pc["valid"], pc["proj_map"], pc["octave"], pc["ldesc"],
                                                               occupied=pc["occupied"], has_obs=pc["has_obs"])),
      "cpu", us(lambda: tlp.oracle_map(ora, pc, False, 0.8), 10), flush=True)
