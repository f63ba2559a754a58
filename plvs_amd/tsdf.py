"""Host-side mirror of PLVS's chisel volumetric back end.

  PointCloudMapChisel.InsertCloud(cloud_camera, Twc, max_range)
      <-> PLVS2::PointCloudMapChisel<PointT>::InsertCloud  (src/PointCloudMapChisel.cc:76-98)
          -> ChiselServer::SetPointCloud / IntegrateLastPointCloud
             (Thirdparty/chisel_server/src/ChiselServer.cpp:561, 664)
          -> chisel::Chisel::IntegratePointCloudWidthDepth
             (Thirdparty/open_chisel/src/Chisel.cpp:442-585)
  Clear() <-> PointCloudMap::Clear ; GetChunk/ChunkIds give the map back.

All compute happens in libplvs_hip.so; this file only marshals arguments.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class DepthBatch(ctypes.Structure):
    """plvs_depth_batch (include/plvs_hip.h)."""
    _fields_ = [("d_depth", ctypes.c_void_p), ("d_bgr", ctypes.c_void_p),
                ("depth_image_stride", ctypes.c_size_t), ("bgr_image_stride", ctypes.c_size_t),
                ("depth_pitch", ctypes.c_int), ("bgr_pitch", ctypes.c_int),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("step", ctypes.c_int),
                ("d_grid_points", ctypes.c_void_p), ("min_depth", ctypes.c_double), ("max_depth", ctypes.c_double),
                ("d_kfid", ctypes.c_void_p)]


class TsdfChisel:
    """Thin RAII wrapper of the plvs_hip_tsdf_chisel_* C ABI."""

    def __init__(self, resolution, max_chunks=None, shard_rank=0, shard_count=1, order_free=False):
        """order_free=False: bit-identical to the reference's sequential loop.  True: the visits
        of a call are summed per voxel and applied at once (float-rounding tolerance on sdf /
        weight; kfid and colour stay exact)."""
        p = _lib.TsdfChiselParams()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_default_params(ctypes.c_float(resolution), ctypes.byref(p)))
        if max_chunks is not None:
            p.max_chunks = int(max_chunks)
        p.shard_rank, p.shard_count = int(shard_rank), int(shard_count)
        p.order_free = 1 if order_free else 0
        self.params = p
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_create(ctypes.byref(p), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib.plvs_hip_tsdf_chisel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_clear(self._h))

    def integrate(self, xyz, rgb, kfid, Twc):
        """Host flavour: numpy arrays in camera frame + 3x4 pose."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
        kfid = None if kfid is None else np.ascontiguousarray(kfid, dtype=np.uint32)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_integrate(
            self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgb), _lib.np_ptr(kfid), xyz.shape[0], _lib.np_ptr(Twc)))

    def queue(self, xyz, rgb, kfid, Twc):
        """Host flavour, deferred: the cloud is uploaded and integrated by the next flush() — or by whatever reads or
        changes the map first — together with everything else queued, in ONE call of the batch pipeline."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
        kfid = None if kfid is None else np.ascontiguousarray(kfid, dtype=np.uint32)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = _lib.lib.plvs_hip_tsdf_chisel_queue
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgb), _lib.np_ptr(kfid), xyz.shape[0], _lib.np_ptr(Twc)))

    def flush(self):
        f = _lib.lib.plvs_hip_tsdf_chisel_flush
        f.argtypes = [ctypes.c_void_p]
        _lib.check(f(self._h))

    def queued(self):
        n = ctypes.c_int()
        f = _lib.lib.plvs_hip_tsdf_chisel_queued
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, ctypes.byref(n)))
        return n.value

    def carve(self, depth, fx, fy, cx, cy, Twc, near=0.05, far=5.0, carving_dist=0.05):
        """Depth-image carving (Chisel.cpp:394-438 / ProjectionIntegrator::CarveWithDepth); depth is a
        float32 image (NaN = no measurement).  Returns the number of carved chunks."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        n = ctypes.c_int()
        f = _lib.lib.plvs_hip_tsdf_chisel_carve
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_float] * 6 + \
                     [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(depth), depth.shape[1], depth.shape[0], fx, fy, cx, cy, near, far,
                     _lib.np_ptr(Twc), carving_dist, ctypes.byref(n)))
        return n.value

    def mesh_chunks(self, chunk_ids, halo_ok=False):
        """ChunkManager::RecomputeMesh for every chunk id of the list ([n,3] ints, e.g. the 27-neighbourhood of
        updated_chunk_ids() = Chisel's meshesToUpdate).  -> dict(vertices, normals, colors [m,3] f32, kfids [m] u32,
        chunk_first [n+1]): chunk c owns rows chunk_first[c]:chunk_first[c+1].  Sharded map: None (with halo_ok) when
        chunks of other ranks are needed first — halo_missing() lists them (plvs_amd.shard.sharded_mesh_chunks)."""
        ids = np.ascontiguousarray(chunk_ids, dtype=np.int32).reshape(-1, 3)
        first = np.zeros(ids.shape[0] + 1, np.int32)
        n = ctypes.c_int()
        f = _lib.lib.plvs_hip_tsdf_chisel_mesh_chunks
        f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 2
        cap = 0
        while True:
            v = np.zeros((max(cap, 1), 3), np.float32)
            nr = np.zeros((max(cap, 1), 3), np.float32)
            c = np.zeros((max(cap, 1), 3), np.float32)
            k = np.zeros(max(cap, 1), np.uint32)
            rc = f(self._h, _lib.np_ptr(ids), ids.shape[0], _lib.np_ptr(v), _lib.np_ptr(nr), _lib.np_ptr(c),
                   _lib.np_ptr(k), cap, _lib.np_ptr(first), ctypes.byref(n))
            if rc == _lib.PLVS_ERR_CAPACITY and n.value > cap:
                cap = n.value            # first call sizes the mesh, second call fills it
                continue
            if rc == _lib.PLVS_ERR_HALO and halo_ok:
                return None
            _lib.check(rc)
            m = n.value
            return dict(vertices=v[:m], normals=nr[:m], colors=c[:m], kfids=k[:m], chunk_first=first)

    def set_chunk(self, cx, cy, cz, sdf, weight, kfid, rgbw):
        """Creates or replaces one chunk (4096 voxels, id = (z * 16 + y) * 16 + x): the counterpart of get_chunk."""
        a = [np.ascontiguousarray(sdf, np.float32).reshape(4096), np.ascontiguousarray(weight, np.float32).reshape(4096),
             np.ascontiguousarray(kfid, np.uint32).reshape(4096), np.ascontiguousarray(rgbw, np.uint32).reshape(4096)]
        f = _lib.lib.plvs_hip_tsdf_chisel_upload_chunk
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
        _lib.check(f(self._h, int(cx), int(cy), int(cz), *[_lib.np_ptr(x) for x in a]))

    # ---- Chisel::Deform (include/plvs_hip.h: plvs_hip_tsdf_chisel_enable_deform / _deform / _deform_mesh)
    def enable_deform(self):
        """On the EMPTY map: from here on the map keeps the reference's chunk-container order (one cloud per integrate
        call)."""
        f = _lib.lib.plvs_hip_tsdf_chisel_enable_deform
        f.argtypes = [ctypes.c_void_p]
        _lib.check(f(self._h))
        return self

    def chunk_order(self):
        """Chunk ids in the iteration order of the reference's std::unordered_map (deform enabled)."""
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), np.int32)
        m = ctypes.c_int()
        f = _lib.lib.plvs_hip_tsdf_chisel_chunk_order
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(ids), n, ctypes.byref(m)))
        return ids[:m.value]

    class _DeformStats(ctypes.Structure):
        _fields_ = [("new_chunks", ctypes.c_int32), ("moved", ctypes.c_int64), ("discarded", ctypes.c_int64),
                    ("undefined", ctypes.c_int64)]

    def deform(self, kfids, Rt):
        """Chisel::Deform.  kfids [n] strictly increasing, Rt [n, 12] (R row-major, then t).
        -> dict(new_chunks, moved, discarded, undefined)"""
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        st = self._DeformStats()
        f = _lib.lib.plvs_hip_tsdf_chisel_deform
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(kfids), _lib.np_ptr(Rt), len(kfids), ctypes.byref(st)))
        return dict(new_chunks=st.new_chunks, moved=st.moved, discarded=st.discarded, undefined=st.undefined)

    @staticmethod
    def deform_mesh(vertices, normals, vertex_kfid, kfids, Rt):
        """The mesh half of ChunkManager::Deform: -> (vertices, normals) moved by their key frame's transformation."""
        v = np.ascontiguousarray(vertices, np.float32).copy()
        nr = np.ascontiguousarray(normals, np.float32).copy()
        vk = np.ascontiguousarray(vertex_kfid, np.uint32)
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        f = _lib.lib.plvs_hip_tsdf_chisel_deform_mesh
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_int]
        _lib.check(f(_lib.np_ptr(v), _lib.np_ptr(nr), _lib.np_ptr(vk), len(vk), _lib.np_ptr(kfids), _lib.np_ptr(Rt), len(kfids)))
        return v, nr

    # ---- halo of a sharded map, for meshing (include/plvs_hip.h: plvs_hip_tsdf_chisel_halo_*)
    HALO_WORDS = 4 * 4096     # a chunk on the wire: sdf, weight, kfid, rgbw planes

    def mesh_probe(self, chunk_ids):
        """The stages of mesh_chunks on the device only -> the number of foreign chunks the list's meshes reach for and
        this rank does not hold (halo_missing() lists them); 0: mesh_chunks will succeed."""
        ids = np.ascontiguousarray(chunk_ids, dtype=np.int32).reshape(-1, 3)
        f = _lib.lib.plvs_hip_tsdf_chisel_mesh_probe
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        n = ctypes.c_int()
        _lib.check(f(self._h, _lib.np_ptr(ids), ids.shape[0], ctypes.byref(n)))
        return n.value

    def halo_gather(self, comm, chunk_ids):
        """The whole halo exchange behind the C ABI over an ncclComm_t (collective) -> chunks imported."""
        ids = np.ascontiguousarray(chunk_ids, dtype=np.int32).reshape(-1, 3)
        f = _lib.lib.plvs_hip_tsdf_chisel_halo_gather
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        n = ctypes.c_int()
        _lib.check(f(self._h, comm, _lib.np_ptr(ids), ids.shape[0], ctypes.byref(n), _lib.current_stream_ptr()))
        return n.value

    def halo_missing(self):
        """[k,3] int32: the chunks of other ranks the last mesh_chunks call looked for and did not hold."""
        f = _lib.lib.plvs_hip_tsdf_chisel_halo_missing
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        n = ctypes.c_int()
        rc = f(self._h, None, 0, ctypes.byref(n))
        if rc != _lib.PLVS_ERR_CAPACITY:
            _lib.check(rc)
        ids = np.zeros((max(n.value, 1), 3), np.int32)
        if n.value:
            _lib.check(f(self._h, _lib.np_ptr(ids), n.value, ctypes.byref(n)))
        return ids[:n.value]

    def halo_lookup(self, d_ids, d_found):
        """Owner side: d_ids [n,3] int32 (cuda) -> d_found [n] int32 (1: this rank has the chunk)."""
        f = _lib.lib.plvs_hip_tsdf_chisel_halo_lookup
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), int(d_ids.shape[0]), _lib.t_ptr(d_found), _lib.current_stream_ptr()))

    def halo_export(self, d_ids, d_found, d_payload):
        """Owner side: the found chunks, one row of HALO_WORDS int32 each, in request order -> d_payload [nfound, ...]."""
        f = _lib.lib.plvs_hip_tsdf_chisel_halo_export
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), _lib.t_ptr(d_found), int(d_ids.shape[0]), _lib.t_ptr(d_payload),
                     _lib.current_stream_ptr()))

    def halo_import(self, d_ids, d_found, d_payload):
        """Requester side: the ids asked for, the owners' found flags, the payload rows of the found ones."""
        f = _lib.lib.plvs_hip_tsdf_chisel_halo_import
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), _lib.t_ptr(d_found), _lib.t_ptr(d_payload), int(d_ids.shape[0]),
                     int(d_payload.shape[0]), _lib.current_stream_ptr()))

    def halo_clear(self):
        _lib.lib.plvs_hip_tsdf_chisel_halo_clear.argtypes = [ctypes.c_void_p]
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_halo_clear(self._h))

    def integrate_world_normals(self, xyz, rgb, kfid, normals, Twc=None):
        """Chisel::IntegrateWorldPointCloudWithNormals (the LoadMap path): a cloud with normals, Twc identity by default."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        kfid = None if kfid is None else np.ascontiguousarray(kfid, dtype=np.uint32)
        Twc = np.ascontiguousarray(np.eye(4, dtype=np.float32)[:3] if Twc is None else Twc, dtype=np.float32).reshape(3, 4)
        f = _lib.lib.plvs_hip_tsdf_chisel_integrate_world_normals
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgb), _lib.np_ptr(kfid), _lib.np_ptr(normals), xyz.shape[0],
                     _lib.np_ptr(Twc)))

    def integrate_batch_dev(self, d_xyz, d_rgb, d_kfid, offsets, d_Twc):
        """Device flavour: concatenated clouds resident in HBM (torch tensors),
        `offsets` a host int32 array of nclouds+1, d_Twc [nclouds,3,4] f32."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_integrate_batch_dev(
            self._h, _lib.t_ptr(d_xyz), _lib.t_ptr(d_rgb), _lib.t_ptr(d_kfid), _lib.np_ptr(offsets),
            offsets.shape[0] - 1, _lib.t_ptr(d_Twc), _lib.current_stream_ptr()))

    def integrate_depth_batch_dev(self, d_depth, d_bgr, d_grid, step, min_depth, max_depth, d_kfid, d_Twc):
        """GeneratePointCloudInCameraFrameBGRA + InsertCloud in one call (plvs_hip_tsdf_chisel_integrate_depth_batch_dev):
        d_depth [n, h, w] f32, d_bgr [n, h, w, 3] u8, d_grid [ceil(h / step) * ceil(w / step), 2] f32 (InitCamGridPoints),
        d_kfid [n] int32 / uint32 (or None), d_Twc [n, 3, 4] f32 — torch tensors in HBM.  Image rows may be strided."""
        n, hgt, wid = d_depth.shape
        assert d_depth.dtype == torch.float32 and d_depth.stride(2) == 1
        assert d_bgr.dtype == torch.uint8 and tuple(d_bgr.shape) == (n, hgt, wid, 3) and d_bgr.stride(3) == 1 and d_bgr.stride(2) == 3
        assert d_grid.dtype == torch.float32 and d_grid.is_contiguous()
        assert d_kfid is None or (d_kfid.dtype in (torch.int32, torch.uint32) and d_kfid.is_contiguous())
        b = DepthBatch()
        b.d_depth, b.d_bgr = d_depth.data_ptr(), d_bgr.data_ptr()
        b.depth_image_stride = d_depth.stride(0) if n > 1 else hgt * d_depth.stride(1)
        b.bgr_image_stride = d_bgr.stride(0) if n > 1 else hgt * d_bgr.stride(1)
        b.depth_pitch, b.bgr_pitch = d_depth.stride(1), d_bgr.stride(1)
        b.width, b.height, b.step = wid, hgt, int(step)
        b.d_grid_points = d_grid.data_ptr()
        b.min_depth, b.max_depth = float(min_depth), float(max_depth)
        b.d_kfid = d_kfid.data_ptr() if d_kfid is not None else None
        f = _lib.lib.plvs_hip_tsdf_chisel_integrate_depth_batch_dev
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(DepthBatch), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, ctypes.byref(b), n, _lib.t_ptr(d_Twc), _lib.current_stream_ptr()))

    # ---- ray-sharded multi-GPU integrate (order_free, shard_count > 1): walk -> pack -> exchange -> apply
    def shard_walk(self, d_xyz, offsets, d_Twc):
        """Phase 1: this rank walks its tiles of the point stream (tile t belongs to rank t % shard_count).
        Returns the int64 array [shard_count, 3] of (descriptors, voxel sums, colour-run records) bound for every rank."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        counts = np.zeros((self.params.shard_count, 3), np.int64)
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_walk
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
        _lib.check(f(self._h, _lib.t_ptr(d_xyz), _lib.np_ptr(offsets), offsets.shape[0] - 1, _lib.t_ptr(d_Twc),
                     _lib.np_ptr(counts), _lib.current_stream_ptr()))
        return counts

    def shard_pack(self, d_seg, d_rec, d_run):
        """Phase 2: fills the send buffers (torch int32 tensors [descriptors, 8], [sums, 8] and [run records, 6]),
        each grouped by destination rank in rank order."""
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_pack
        f.argtypes = [ctypes.c_void_p] * 5
        _lib.check(f(self._h, _lib.t_ptr(d_seg), _lib.t_ptr(d_rec), _lib.t_ptr(d_run), _lib.current_stream_ptr()))

    def shard_apply(self, d_seg, d_rec, d_run, recv_counts, d_rgb, d_kfid):
        """Phase 3: the received buffers (grouped by source rank in rank order; recv_counts [shard_count, 3]) are
        applied to this rank's chunks, colours through the received runs."""
        recv_counts = np.ascontiguousarray(recv_counts, dtype=np.int64)
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_apply
        f.argtypes = [ctypes.c_void_p] * 8
        _lib.check(f(self._h, _lib.t_ptr(d_seg), _lib.t_ptr(d_rec), _lib.t_ptr(d_run), _lib.np_ptr(recv_counts),
                     _lib.t_ptr(d_rgb), _lib.t_ptr(d_kfid), _lib.current_stream_ptr()))

    def shard_saturated(self):
        """Voxels of this rank whose colour weight reached 254 in the last shard_apply: int32 tensor [n, 4]
        (chunk x, y, z, voxel) on the device — to be all-gathered and noted on every rank."""
        n = ctypes.c_int()
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_saturated
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = f(self._h, None, 0, ctypes.byref(n), _lib.current_stream_ptr())
        if rc == _lib.PLVS_OK:
            return torch.zeros((0, 4), dtype=torch.int32, device="cuda")
        out = torch.zeros((n.value, 4), dtype=torch.int32, device="cuda")
        _lib.check(f(self._h, _lib.t_ptr(out), n.value, ctypes.byref(n), _lib.current_stream_ptr()))
        return out

    def shard_note_saturated(self, d_voxels):
        """Notes voxels other ranks (or this one) reported saturated: this rank's walks stop sending their runs."""
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_note_saturated
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_voxels), int(d_voxels.shape[0]), _lib.current_stream_ptr()))

    def shard_saturated_message(self, d_msg, rows):
        """This rank's message for the saturation all-gather (int32 tensor [rows + 1, 4]): up to `rows` waiting
        voxels, their number in the last row; the rest waits with the handle for the next step."""
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_saturated_message
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_msg), int(rows), _lib.current_stream_ptr()))

    def shard_note_gathered(self, d_gathered, nranks, rows):
        """Notes the all-gathered messages of `nranks` ranks ([nranks * (rows + 1), 4]); no host read."""
        f = _lib.lib.plvs_hip_tsdf_chisel_shard_note_gathered
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_gathered), int(nranks), int(rows), _lib.current_stream_ptr()))

    def last_stats(self):
        s = _lib.TsdfStats()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_last_stats(self._h, ctypes.byref(s)))
        return dict(visits=s.visits, points=s.points, new_chunks=s.new_chunks,
                    updated_chunks=s.updated_chunks, voxels=s.voxels, max_run=s.max_run)

    def set_apply_parts(self, part_segments, min_segments):
        """Tuning knob of the order-free apply stage (results do not depend on it): chunks with more than
        `min_segments` segments in a call are applied in parts of `part_segments`."""
        f = _lib.lib.plvs_hip_tsdf_chisel_set_apply_parts
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _lib.check(f(self._h, int(part_segments), int(min_segments)))

    def set_profiling(self, enable=True):
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_set_profiling(self._h, int(bool(enable))))

    def stage_ms(self):
        """{stage name: accumulated ms}, number of integrate calls covered."""
        ms = (ctypes.c_double * 16)()
        n = ctypes.c_int()
        calls = ctypes.c_int64()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_stage_ms(self._h, ms, 16, ctypes.byref(n), ctypes.byref(calls)))
        names = [_lib.lib.plvs_hip_tsdf_chisel_stage_name_of(self._h, i).decode() for i in range(n.value)]
        return {names[i]: ms[i] for i in range(n.value)}, calls.value

    def updated_chunk_ids_dev(self, d_ids):
        """Writes the updated-chunk id triples into the torch int32 tensor d_ids
        [cap,3]; returns the number of updated chunks."""
        n = ctypes.c_int()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_updated_chunk_ids_dev(
            self._h, _lib.t_ptr(d_ids), d_ids.shape[0], ctypes.byref(n), _lib.current_stream_ptr()))
        return n.value

    def num_chunks(self):
        n = ctypes.c_int()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_num_chunks(self._h, ctypes.byref(n)))
        return n.value

    def chunk_ids(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), dtype=np.int32)
        m = ctypes.c_int()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_chunk_ids(self._h, _lib.np_ptr(ids), n, ctypes.byref(m)))
        return ids[:n]

    def updated_chunk_ids(self):
        m = ctypes.c_int()
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_updated_chunk_ids(self._h, None, 0, ctypes.byref(m)))
        ids = np.zeros((max(m.value, 1), 3), dtype=np.int32)
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_updated_chunk_ids(self._h, _lib.np_ptr(ids), m.value, ctypes.byref(m)))
        return ids[:m.value]

    def get_chunk(self, cx, cy, cz):
        sdf = np.empty(4096, np.float32)
        w = np.empty(4096, np.float32)
        kf = np.empty(4096, np.uint32)
        col = np.empty(4096, np.uint32)
        _lib.check(_lib.lib.plvs_hip_tsdf_chisel_download_chunk(
            self._h, int(cx), int(cy), int(cz), _lib.np_ptr(sdf), _lib.np_ptr(w), _lib.np_ptr(kf), _lib.np_ptr(col)))
        return sdf, w, kf, col


class VoxbloxParams(ctypes.Structure):
    _fields_ = [("voxel_size", ctypes.c_float), ("truncation", ctypes.c_float), ("max_weight", ctypes.c_float),
                ("min_ray_length", ctypes.c_float), ("max_ray_length", ctypes.c_float),
                ("voxel_carving", ctypes.c_int32), ("max_blocks", ctypes.c_int32),
                ("shard_rank", ctypes.c_int32), ("shard_count", ctypes.c_int32)]


_L = _lib.lib
_vp, _i = ctypes.c_void_p, ctypes.c_int
_L.plvs_hip_tsdf_voxblox_default_params.argtypes = [ctypes.c_float, _i, ctypes.POINTER(VoxbloxParams)]
_L.plvs_hip_tsdf_voxblox_create.argtypes = [ctypes.POINTER(VoxbloxParams), ctypes.POINTER(_vp)]
_L.plvs_hip_tsdf_voxblox_destroy.argtypes = [_vp]
_L.plvs_hip_tsdf_voxblox_clear.argtypes = [_vp]
_L.plvs_hip_tsdf_voxblox_integrate.argtypes = [_vp, _vp, _vp, _i, _vp]
_L.plvs_hip_tsdf_voxblox_integrate_batch_dev.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _vp]
_L.plvs_hip_tsdf_voxblox_last_stats.argtypes = [_vp, ctypes.POINTER(_lib.TsdfStats)]
_L.plvs_hip_tsdf_voxblox_num_blocks.argtypes = [_vp, ctypes.POINTER(_i)]
_L.plvs_hip_tsdf_voxblox_block_ids.argtypes = [_vp, _vp, _i, ctypes.POINTER(_i)]
_L.plvs_hip_tsdf_voxblox_updated_block_ids_dev.argtypes = [_vp, _vp, _i, ctypes.POINTER(_i), _vp]
_L.plvs_hip_tsdf_voxblox_download_block.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp]


class TsdfVoxblox:
    """Thin RAII wrapper of the plvs_hip_tsdf_voxblox_* C ABI."""

    def __init__(self, voxel_size, use_carving=False, max_blocks=None, shard_rank=0, shard_count=1,
                 max_ray_length=None, max_weight=None):
        p = VoxbloxParams()
        _lib.check(_L.plvs_hip_tsdf_voxblox_default_params(ctypes.c_float(voxel_size), int(use_carving), ctypes.byref(p)))
        if max_blocks is not None:
            p.max_blocks = int(max_blocks)
        if max_ray_length is not None:
            p.max_ray_length = float(max_ray_length)
        if max_weight is not None:          # TsdfIntegratorBase::Config::max_weight (10000 unless the caller says otherwise)
            p.max_weight = float(max_weight)
        p.shard_rank, p.shard_count = int(shard_rank), int(shard_count)
        self.shard_rank, self.shard_count = int(shard_rank), int(shard_count)
        self.params = p
        self._h = _vp()
        _lib.check(_L.plvs_hip_tsdf_voxblox_create(ctypes.byref(p), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _L.plvs_hip_tsdf_voxblox_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _lib.check(_L.plvs_hip_tsdf_voxblox_clear(self._h))

    def set_deferred_world_blocks(self, enable):
        """plvs_hip_tsdf_voxblox_set_deferred_world_blocks: blocks created by integrate_world_normals stay out of the
        block list / meshes / updated list until the next camera-ray integrate (the reference's behaviour)."""
        f = _L.plvs_hip_tsdf_voxblox_set_deferred_world_blocks
        f.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.check(f(self._h, int(enable)))

    def integrate(self, xyz, rgba, Twc):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        _lib.check(_L.plvs_hip_tsdf_voxblox_integrate(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgba), xyz.shape[0],
                                                      _lib.np_ptr(Twc)))

    def queue(self, xyz, rgba, Twc):
        """Upload a key frame's cloud without integrating it (plvs_hip_tsdf_voxblox_queue); flush() — or any call that reads
        the map — integrates what waits as one batch of the simple integrator."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = _L.plvs_hip_tsdf_voxblox_queue
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgba), xyz.shape[0], _lib.np_ptr(Twc)))

    def flush(self):
        f = _L.plvs_hip_tsdf_voxblox_flush
        f.argtypes = [ctypes.c_void_p]
        _lib.check(f(self._h))

    def queued(self):
        n = ctypes.c_int()
        f = _L.plvs_hip_tsdf_voxblox_queued
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, ctypes.byref(n)))
        return n.value

    def integrate_fast(self, xyz, rgba, Twc):
        """FastTsdfIntegrator::integratePointCloud (integration method "fast", one thread; the reference's approximate sets
        word for word, kept from scan to scan)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = _L.plvs_hip_tsdf_voxblox_integrate_fast
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgba), xyz.shape[0], _lib.np_ptr(Twc)))

    def integrate_fast_batch_dev(self, d_xyz, d_rgba, offsets, d_Twc):
        """The same for several clouds in HBM: every cloud is one scan."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        f = _L.plvs_hip_tsdf_voxblox_integrate_fast_batch_dev
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_xyz), _lib.t_ptr(d_rgba), _lib.np_ptr(offsets), offsets.shape[0] - 1,
                     _lib.t_ptr(d_Twc), _lib.current_stream_ptr()))

    def fast_rounds(self):
        n = ctypes.c_int()
        f = _L.plvs_hip_tsdf_voxblox_fast_rounds
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, ctypes.byref(n)))
        return n.value

    def integrate_merged(self, xyz, rgba, Twc):
        """MergedTsdfIntegrator::integratePointCloud (integration method "merged", one thread)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = _L.plvs_hip_tsdf_voxblox_integrate_merged
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgba), xyz.shape[0], _lib.np_ptr(Twc)))

    def integrate_world_normals(self, xyz, rgba, normals, Twc=None):
        """TsdfIntegratorBase::integrateWorlPointCloud (the LoadMap path): a cloud with normals, T identity by default."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        Twc = np.ascontiguousarray(np.eye(4, dtype=np.float32)[:3] if Twc is None else Twc, dtype=np.float32).reshape(3, 4)
        f = _L.plvs_hip_tsdf_voxblox_integrate_world_normals
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.np_ptr(xyz), _lib.np_ptr(rgba), _lib.np_ptr(normals), xyz.shape[0], _lib.np_ptr(Twc)))

    def integrate_batch_dev(self, d_xyz, d_rgba, offsets, d_Twc):
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        _lib.check(_L.plvs_hip_tsdf_voxblox_integrate_batch_dev(
            self._h, _lib.t_ptr(d_xyz), _lib.t_ptr(d_rgba), _lib.np_ptr(offsets), offsets.shape[0] - 1,
            _lib.t_ptr(d_Twc), _lib.current_stream_ptr()))

    # ---- ray-sharded multi-GPU integrate ("simple", shard_count >= 1): walk -> pack -> exchange -> apply
    def shard_walk(self, d_xyz, offsets, d_Twc):
        """Phase 1: this rank casts the rays of its clouds (cloud c belongs to rank c % shard_count).  Returns the
        int64 array [shard_count] of visit records bound for every rank."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        counts = np.zeros(max(1, self.shard_count), np.int64)
        f = _L.plvs_hip_tsdf_voxblox_shard_walk
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
        _lib.check(f(self._h, _lib.t_ptr(d_xyz), _lib.np_ptr(offsets), offsets.shape[0] - 1, _lib.t_ptr(d_Twc),
                     _lib.np_ptr(counts), _lib.current_stream_ptr()))
        return counts

    def shard_pack(self, d_send):
        """Phase 2: fills the send buffer (torch int32 tensor [records, 4]), grouped by destination rank in rank order."""
        f = _L.plvs_hip_tsdf_voxblox_shard_pack
        f.argtypes = [ctypes.c_void_p] * 3
        _lib.check(f(self._h, _lib.t_ptr(d_send), _lib.current_stream_ptr()))

    def shard_apply(self, d_recv, recv_counts, d_xyz, d_rgba, offsets, d_Twc):
        """Phase 3: the received records (grouped by source rank in rank order; recv_counts [shard_count]) are applied
        to this rank's blocks in the reference's order; the clouds again: the owner recomputes a visit's operands."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        recv_counts = np.ascontiguousarray(recv_counts, dtype=np.int64)
        f = _L.plvs_hip_tsdf_voxblox_shard_apply
        f.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 2
        _lib.check(f(self._h, _lib.t_ptr(d_recv), _lib.np_ptr(recv_counts), _lib.t_ptr(d_xyz), _lib.t_ptr(d_rgba),
                     _lib.np_ptr(offsets), offsets.shape[0] - 1, _lib.t_ptr(d_Twc), _lib.current_stream_ptr()))

    def last_stats(self):
        s = _lib.TsdfStats()
        _lib.check(_L.plvs_hip_tsdf_voxblox_last_stats(self._h, ctypes.byref(s)))
        return dict(visits=s.visits, points=s.points, new_chunks=s.new_chunks,
                    updated_chunks=s.updated_chunks, voxels=s.voxels, max_run=s.max_run)

    def num_chunks(self):
        n = _i()
        _lib.check(_L.plvs_hip_tsdf_voxblox_num_blocks(self._h, ctypes.byref(n)))
        return n.value

    def chunk_ids(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), dtype=np.int32)
        m = _i()
        _lib.check(_L.plvs_hip_tsdf_voxblox_block_ids(self._h, _lib.np_ptr(ids), n, ctypes.byref(m)))
        return ids[:n]

    def updated_chunk_ids_dev(self, d_ids):
        n = _i()
        _lib.check(_L.plvs_hip_tsdf_voxblox_updated_block_ids_dev(
            self._h, _lib.t_ptr(d_ids), d_ids.shape[0], ctypes.byref(n), _lib.current_stream_ptr()))
        return n.value

    def updated_chunk_ids(self):
        """[n,3] ids of the blocks the last integrate call visited (Block::updated())."""
        n = _i()
        f = _L.plvs_hip_tsdf_voxblox_updated_block_ids
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, None, 0, ctypes.byref(n)))
        ids = np.zeros((max(n.value, 1), 3), np.int32)
        _lib.check(f(self._h, _lib.np_ptr(ids), n.value, ctypes.byref(n)))
        return ids[:n.value]

    def mesh_blocks(self, block_ids):
        """MeshIntegrator::updateMeshForBlock for every block id of the list ([n,3] ints).  -> dict(vertices, normals
        [m,3] f32, colors [m,4] u8 (r, g, b, a), block_first [n+1]): block c owns rows block_first[c]:block_first[c+1]."""
        ids = np.ascontiguousarray(block_ids, dtype=np.int32).reshape(-1, 3)
        first = np.zeros(ids.shape[0] + 1, np.int32)
        n = ctypes.c_int()
        f = _L.plvs_hip_tsdf_voxblox_mesh_blocks
        f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2
        cap = 0
        while True:
            v = np.zeros((max(cap, 1), 3), np.float32)
            nr = np.zeros((max(cap, 1), 3), np.float32)
            c = np.zeros((max(cap, 1), 4), np.uint8)
            rc = f(self._h, _lib.np_ptr(ids), ids.shape[0], _lib.np_ptr(v), _lib.np_ptr(nr), _lib.np_ptr(c), cap,
                   _lib.np_ptr(first), ctypes.byref(n))
            if rc == _lib.PLVS_ERR_CAPACITY and n.value > cap:
                cap = n.value            # first call sizes the mesh, second call fills it
                continue
            _lib.check(rc)
            m = n.value
            return dict(vertices=v[:m], normals=nr[:m], colors=c[:m], block_first=first)

    def set_chunk(self, bx, by, bz, distance, weight, rgba):
        """Creates or replaces one block (4096 voxels, index x + 16 * (y + 16 * z)): Layer::addBlockFromProto(kReplace)."""
        a = [np.ascontiguousarray(distance, np.float32).reshape(4096), np.ascontiguousarray(weight, np.float32).reshape(4096),
             np.ascontiguousarray(rgba, np.uint32).reshape(4096)]
        f = _L.plvs_hip_tsdf_voxblox_upload_block
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
        _lib.check(f(self._h, int(bx), int(by), int(bz), *[_lib.np_ptr(x) for x in a]))

    # ---- halo of a sharded map, for meshing (include/plvs_hip.h: plvs_hip_tsdf_voxblox_halo_*)
    HALO_WORDS = 3 * 4096     # a block on the wire: distance, weight, rgba planes

    def halo_lookup(self, d_ids, d_found):
        f = _L.plvs_hip_tsdf_voxblox_halo_lookup
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), int(d_ids.shape[0]), _lib.t_ptr(d_found), _lib.current_stream_ptr()))

    def halo_export(self, d_ids, d_found, d_payload):
        f = _L.plvs_hip_tsdf_voxblox_halo_export
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), _lib.t_ptr(d_found), int(d_ids.shape[0]), _lib.t_ptr(d_payload),
                     _lib.current_stream_ptr()))

    def halo_import(self, d_ids, d_found, d_payload):
        f = _L.plvs_hip_tsdf_voxblox_halo_import
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.check(f(self._h, _lib.t_ptr(d_ids), _lib.t_ptr(d_found), _lib.t_ptr(d_payload), int(d_ids.shape[0]),
                     int(d_payload.shape[0]), _lib.current_stream_ptr()))

    def halo_gather(self, comm, block_ids):
        """The halo exchange behind the C ABI over an ncclComm_t (collective) -> blocks imported."""
        ids = np.ascontiguousarray(block_ids, dtype=np.int32).reshape(-1, 3)
        f = _L.plvs_hip_tsdf_voxblox_halo_gather
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        n = ctypes.c_int()
        _lib.check(f(self._h, comm, _lib.np_ptr(ids), ids.shape[0], ctypes.byref(n), _lib.current_stream_ptr()))
        return n.value

    def halo_clear(self):
        _L.plvs_hip_tsdf_voxblox_halo_clear.argtypes = [ctypes.c_void_p]
        _lib.check(_L.plvs_hip_tsdf_voxblox_halo_clear(self._h))

    def get_chunk(self, bx, by, bz):
        d = np.empty(4096, np.float32)
        w = np.empty(4096, np.float32)
        c = np.empty(4096, np.uint32)
        _lib.check(_L.plvs_hip_tsdf_voxblox_download_block(self._h, int(bx), int(by), int(bz), _lib.np_ptr(d),
                                                           _lib.np_ptr(w), _lib.np_ptr(c)))
        return d, w, c


class PointCloudMapVoxblox:
    """Same surface as PLVS2::PointCloudMapVoxblox for the integrate path
    (src/PointCloudMapVoxblox.cc:48-99)."""
    # "simple", "merged" and "fast" reproduce the reference's one-thread schedules bit for bit.
    # The default is the reference's own static default, "fast" (src/PointCloudMapVoxblox.cc:44; its YAMLs set it too):
    # a drop-in gives the maps a PLVS run gives.  "fast" is no speed-up on the device — it skips nine tenths of the
    # updates at the price of several rounds over the scan — so "simple" (the integrator of the measured configs[3]
    # leg and of the sharded path) is the documented opt-in: integration_method="simple", or a subclass attribute.
    skIntegrationMethod = "fast"

    def __init__(self, resolution, use_carving=False, max_blocks=None, queue_insertions=True, integration_method=None):
        if integration_method is not None:
            self.skIntegrationMethod = integration_method
        if self.skIntegrationMethod not in ("simple", "merged", "fast"):
            raise ValueError(f"unknown voxblox integration method {self.skIntegrationMethod!r}")
        self._tsdf = TsdfVoxblox(resolution, use_carving, max_blocks)
        # "simple": InsertCloud uploads, UpdateMap (the one reader of the layer) integrates what waits as one batch
        self._queue = bool(queue_insertions) and self.skIntegrationMethod == "simple"
        self._pending = False
        self._updated = set()        # the blocks whose updated() flag is set
        self.mesh_layer = {}         # block id -> dict(vertices, normals, colors): voxblox::MeshLayer

    def InsertCloud(self, cloud_camera, Twc, max_range=None):
        print("PointCloudMapVoxblox<PointT>::InsertCloud()")
        Twc = np.asarray(Twc, dtype=np.float32)[:3, :4]
        if self.skIntegrationMethod == "merged":
            self._tsdf.integrate_merged(cloud_camera["xyz"], cloud_camera["rgba"], Twc)
        elif self.skIntegrationMethod == "fast":
            self._tsdf.integrate_fast(cloud_camera["xyz"], cloud_camera["rgba"], Twc)
        elif self._queue:
            self._tsdf.queue(cloud_camera["xyz"], cloud_camera["rgba"], Twc)
            self._pending = True
            return
        else:
            self._tsdf.integrate(cloud_camera["xyz"], cloud_camera["rgba"], Twc)
        self._mark_updated()

    def _mark_updated(self):
        for b in self._tsdf.updated_chunk_ids():          # tsdf_integrator.cc:151
            self._updated.add((int(b[0]), int(b[1]), int(b[2])))

    def _flush(self):
        if self._pending:
            self._tsdf.flush()
            self._pending = False
            self._mark_updated()

    def SetReferenceLoadMapVisibility(self, on):
        """The reference's LoadMap leaves the loaded cloud's blocks outside the layer until the next InsertCloud
        (integrateWorlPointCloud never publishes them); on = reproduce that, off (default) = the map shows at once."""
        self._flush()
        self._tsdf.set_deferred_world_blocks(on)

    def LoadMap(self, cloud):
        """LoadMap of a saved cloud once PointCloudMap::LoadMap has read it (src/PointCloudMapVoxblox.cc:233-258):
        TsdfServer::insertWorldPointCloud(cloud, identity) — every point along its normal — then UpdateMap.  (The
        `.proto` volume file of TsdfServer::loadMap is protobuf I/O on the host: not mirrored.)"""
        self._flush()
        xyz = np.stack([cloud["x"], cloud["y"], cloud["z"]], -1)
        rgba = np.stack([cloud["r"], cloud["g"], cloud["b"], cloud["a"]], -1)
        self._tsdf.integrate_world_normals(xyz, rgba, cloud["normal"])
        for b in self._tsdf.updated_chunk_ids():
            self._updated.add((int(b[0]), int(b[1]), int(b[2])))
        return self.UpdateMap()

    def SaveLayer(self):
        """TsdfServer::saveMap's device side (tsdf_server.cc:859-863, io::SaveLayer): every block's voxel planes, as
        {block id: (distance, weight, rgba)} — the `.proto` serialisation itself stays with the reference's protobuf code."""
        self._flush()
        return {tuple(int(v) for v in b): self._tsdf.get_chunk(*b) for b in self._tsdf.chunk_ids()}

    def LoadLayer(self, blocks):
        """TsdfServer::loadMap's device side (tsdf_server.cc:865-872: LoadBlocksFromFile with kReplace): every stored block
        replaces / creates its block and is marked updated (core/layer_inl.h:195-197, :215).  No UpdateMap: the reference's
        `.proto` branch of LoadMap does not call it either (src/PointCloudMapVoxblox.cc:237-241)."""
        self._flush()
        for bid, (d, w, c) in blocks.items():
            self._tsdf.set_chunk(bid[0], bid[1], bid[2], d, w, c)
            self._updated.add((int(bid[0]), int(bid[1]), int(bid[2])))
        return True

    # colorVoxbloxToMsg / colorMsgToVoxblox (voxblox_ros/conversions.h:44-60): a channel goes through a float in [0, 1]
    _CLOUD_COLOUR = ((np.arange(256) / 255.0).astype(np.float32).astype(np.float64) * 255.0).astype(np.uint8)

    def UpdateMap(self):
        """updateMesh (generateMesh(only_mesh_updated_blocks, clear_updated_flag), tsdf_server.cc:775-787) +
        getMeshAsPointcloud (voxblox_ros/mesh_vis.h:272-318, ColorMode::kColor)  (src/PointCloudMapVoxblox.cc:160-179).
        -> the output cloud as a structured array (x, y, z, normal, r, g, b), meshes walked in block-id order (the
        reference walks its hash map)."""
        self._flush()
        todo = sorted(self._updated)
        if todo:
            m = self._tsdf.mesh_blocks(np.array(todo, np.int32))
            first = m["block_first"]
            for i, bid in enumerate(todo):
                a, b = int(first[i]), int(first[i + 1])
                # allocateMeshPtrByIndex + mesh->clear(): an updated block always owns a (possibly empty) mesh
                self.mesh_layer[bid] = dict(vertices=m["vertices"][a:b].copy(), normals=m["normals"][a:b].copy(),
                                            colors=m["colors"][a:b].copy())
            self._updated.clear()
        from .cloudgen import POINT_SURFEL
        n = sum(len(v["vertices"]) for v in self.mesh_layer.values())
        cloud = np.zeros(n, POINT_SURFEL)
        o = 0
        for bid in sorted(self.mesh_layer):
            v = self.mesh_layer[bid]
            k = len(v["vertices"])
            if not k:
                continue
            cloud["x"][o:o + k], cloud["y"][o:o + k], cloud["z"][o:o + k] = v["vertices"].T
            cloud["normal"][o:o + k] = v["normals"]
            rgb = self._CLOUD_COLOUR[v["colors"][:, :3]]
            cloud["r"][o:o + k], cloud["g"][o:o + k], cloud["b"][o:o + k] = rgb.T
            o += k
        self.point_cloud = cloud
        return cloud

    def Clear(self):
        self._tsdf.clear()      # (drops what was queued)
        self._pending = False
        self._updated.clear()
        self.mesh_layer.clear()

    @property
    def tsdf(self):
        self._flush()
        return self._tsdf


class PointCloudMapChisel:
    """Same surface as PLVS2::PointCloudMapChisel for the integrate path."""

    def __init__(self, resolution, min_depth=0.1, max_depth=5.0, use_carving=False, max_chunks=None,
                 carving_dist=0.05, near_plane_dist=0.05, far_plane_dist=5.0, bResetOnSparseMapChange=True,
                 bCloudDeformationOnSparseMapChange=False, queue_insertions=True):
        # queue_insertions: InsertCloud uploads the key frame's cloud and UpdateMap (or whatever reads or changes the map
        # first) integrates the ones waiting — at most five between two UpdateMap calls of PointCloudMapping — in one
        # batch: the reference only reads the map in UpdateMap (src/PointCloudMapping.cc:540-552, 594-598), and the batch
        # gives the same map bit for bit in the (default) ordered mode.  Off with the deformation bookkeeping, which
        # follows the map call by call.
        self.queue_insertions = queue_insertions and not bCloudDeformationOnSparseMapChange
        self.resolution = resolution
        self.min_depth, self.max_depth = min_depth, max_depth
        self.use_carving, self.carving_dist = use_carving, carving_dist
        self.near_plane_dist, self.far_plane_dist = near_plane_dist, far_plane_dist
        self.bResetOnSparseMapChange = bResetOnSparseMapChange
        self.bCloudDeformationOnSparseMapChange = bCloudDeformationOnSparseMapChange
        self._tsdf = TsdfChisel(resolution, max_chunks=max_chunks)
        if bCloudDeformationOnSparseMapChange:
            self._tsdf.enable_deform()       # the map keeps the reference's chunk order from its first cloud on
        self._meshes_to_update = set()       # Chisel::meshesToUpdate
        self.all_meshes = {}                 # chunk id -> dict(vertices, normals, colors, kfids)
        self._pending = False                # clouds queued since the last _flush (their chunks are not yet marked)

    def InsertCloud(self, cloud_camera, Twc, max_range=None):
        """cloud_camera: dict/obj with xyz [n,3] f32, rgb [n,3] u8 (r,g,b members of
        the pcl point), kfid [n] u32.  Twc: 3x4 (or 4x4) camera pose."""
        print("PointCloudMapChisel<PointT>::InsertCloud()")
        Twc = np.asarray(Twc, dtype=np.float32)[:3, :4]
        if self.queue_insertions:
            self._tsdf.queue(cloud_camera["xyz"], cloud_camera["rgb"], cloud_camera.get("kfid"), Twc)
            self._pending = True
            return
        self._tsdf.integrate(cloud_camera["xyz"], cloud_camera["rgb"], cloud_camera.get("kfid"), Twc)
        self._mark_updated()

    def _mark_updated(self):
        for c in self._tsdf.updated_chunk_ids():          # Chisel.cpp:553-568
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        self._meshes_to_update.add((int(c[0]) + dx, int(c[1]) + dy, int(c[2]) + dz))

    def _flush(self):
        """Integrates the queued key frames (one batch) and notes the meshes they invalidate.  `_pending`, not the
        handle's queue length, decides: a C reader reached in between flushes the queue itself (PLVS_FLUSH_QUEUE), and
        the batch's chunks must still be marked — the handle's updated list stays that batch's until the next integrate."""
        if self._pending:
            if self._tsdf.queued():
                self._tsdf.flush()
            self._pending = False
            self._mark_updated()

    def InsertCloudWithDepth(self, cloud_camera, Twc, depthImage, fx, fy, cx, cy, max_range=None):
        """src/PointCloudMapChisel.cc:100-133: with useCarving the depth image's frustum is carved first
        (Chisel.cpp:394-438), then the cloud is integrated."""
        Twc = np.asarray(Twc, dtype=np.float32)[:3, :4]
        if self.use_carving:
            self._flush()      # (carving reads and changes the map: what is waiting goes in first)
            n = self._tsdf.carve(depthImage, fx, fy, cx, cy, Twc, near=self.near_plane_dist, far=self.far_plane_dist,
                                 carving_dist=self.carving_dist)
            if n:      # meshesToUpdate[chunkID] = true for every carved chunk, the chunk alone (Chisel.cpp:432)
                for c in self._tsdf.updated_chunk_ids():
                    self._meshes_to_update.add((int(c[0]), int(c[1]), int(c[2])))
        self.InsertCloud(cloud_camera, Twc, max_range)

    def UpdateMap(self):
        """UpdateMesh + GetPointCloud (src/PointCloudMapChisel.cc:228-246): -> the output cloud as a structured
        array (x, y, z, normal, r, g, b, kfid), meshes walked in chunk-id order."""
        self._flush()
        todo = sorted(self._meshes_to_update)
        if todo:
            m = self._tsdf.mesh_chunks(np.array(todo, np.int32))
            first = m["chunk_first"]
            for i, cid in enumerate(todo):
                a, b = int(first[i]), int(first[i + 1])
                if b > a:
                    self.all_meshes[cid] = dict(vertices=m["vertices"][a:b].copy(), normals=m["normals"][a:b].copy(),
                                                colors=m["colors"][a:b].copy(), kfids=m["kfids"][a:b].copy())
                elif cid in self.all_meshes:
                    # RecomputeMesh re-uses the mesh object already in allMeshes and GenerateMesh clears it first
                    # (ChunkManager.cpp:581): a chunk whose surface is gone (carved / reset) keeps an EMPTY mesh
                    e = self.all_meshes[cid]
                    self.all_meshes[cid] = dict(vertices=e["vertices"][:0], normals=e["normals"][:0],
                                                colors=e["colors"][:0], kfids=e["kfids"][:0])
            self._meshes_to_update.clear()
        from .cloudgen import POINT_SURFEL
        n = sum(len(v["kfids"]) for v in self.all_meshes.values())
        cloud = np.zeros(n, POINT_SURFEL)
        o = 0
        for cid in sorted(self.all_meshes):
            v = self.all_meshes[cid]
            k = len(v["kfids"])
            c = cloud[o:o + k]
            c["x"], c["y"], c["z"] = v["vertices"][:, 0], v["vertices"][:, 1], v["vertices"][:, 2]
            c["normal"] = v["normals"]
            col = (v["colors"] * np.float32(255)).astype(np.uint8)          # point.r = meshCol[0]*255
            c["r"], c["g"], c["b"] = col[:, 0], col[:, 1], col[:, 2]
            c["kfid"] = v["kfids"]
            o += k
        return cloud

    def LoadMap(self, cloud):
        """LoadMap once PointCloudMap::LoadMap has read the saved cloud (src/PointCloudMapChisel.cc:527-546):
        ChiselServer::IntegrateWorldPointCloud(cloud, identity) — every point along its normal — then UpdateMap.
        cloud: structured POINT_SURFEL array (or a dict with xyz, rgb, kfid, normal)."""
        if isinstance(cloud, dict):
            xyz, rgb, kfid, nrm = cloud["xyz"], cloud["rgb"], cloud.get("kfid"), cloud["normal"]
        else:
            xyz = np.stack([cloud["x"], cloud["y"], cloud["z"]], -1)
            rgb = np.stack([cloud["r"], cloud["g"], cloud["b"]], -1)
            kfid, nrm = cloud["kfid"], cloud["normal"]
        self._flush()
        self._tsdf.integrate_world_normals(xyz, rgb, kfid, nrm)
        for c in self._tsdf.updated_chunk_ids():          # Chisel.cpp:349-365
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        self._meshes_to_update.add((int(c[0]) + dx, int(c[1]) + dy, int(c[2]) + dz))
        return self.UpdateMap()

    def InsertData(self, pData):
        """PointCloudMapInput dispatch (src/PointCloudMapChisel.cc:192-225): only the
        point-cloud input type is handled; anything else terminates the reference."""
        t = pData.get("type", "kPointCloud")
        if t != "kPointCloud":
            raise SystemExit(-1)
        self.InsertCloud(pData["pPointCloud"], pData["Twc"], pData.get("maxRange", self.max_depth))

    def OnMapChange(self, mapKfidToRt=None):
        """src/PointCloudMapChisel.cc:262-274 / :389-496.  With bResetOnSparseMapChange: ChiselServer::Reset.  With
        bCloudDeformationOnSparseMapChange: UpdateMap, ChiselServer::Deform(mapKfidToRt), UpdateMap.  mapKfidToRt:
        {kfid: (R 3x3, t 3)} — Twc_new * Tcw_at_integration of every valid key frame, which the caller derives from its
        key frames as :420-478 does.  -> the output cloud after the change."""
        if self.bResetOnSparseMapChange:
            self._tsdf.clear()                               # Chisel::Reset: chunks, meshes, meshesToUpdate
            self._pending = False                            # (what was queued went with the map)
            self._meshes_to_update.clear()
            self.all_meshes.clear()
        if self.bCloudDeformationOnSparseMapChange:
            self.UpdateMap()
            kfids = np.array(sorted(mapKfidToRt or {}), np.uint32)
            Rt = np.zeros((len(kfids), 12), np.float32)
            for i, k in enumerate(kfids):
                R, t = mapKfidToRt[int(k)]
                Rt[i, :9] = np.asarray(R, np.float32).reshape(9)
                Rt[i, 9:] = np.asarray(t, np.float32).reshape(3)
            self._tsdf.deform(kfids, Rt)
            for cid, m in self.all_meshes.items():           # ChunkManager.cpp:1020-1051: the stored meshes move too
                if len(m["kfids"]):
                    m["vertices"], m["normals"] = TsdfChisel.deform_mesh(m["vertices"], m["normals"], m["kfids"], kfids, Rt)
        return self.UpdateMap()

    def Clear(self):
        self._tsdf.clear()
        self._pending = False
        self._meshes_to_update.clear()
        self.all_meshes.clear()

    @property
    def tsdf(self):
        self._flush()
        return self._tsdf
