"""Multi-GPU sharding of the voxel-block hash (new design — the
reference is single-process).

owner(block) = three-prime spatial hash(block id) mod world_size, the very hash
both reference back ends key their block maps with (ChunkHasher,
open_chisel ChunkManager.h:42-54; AnyIndexHash, voxblox block_hash.h:15-26).
Every rank receives the same clouds, ray-casts all points and applies only the
visits of the blocks it owns (plvs_tsdf_*_params.shard_rank / shard_count), so
a voxel's updates stay in reference order on its single owner.  The one
exchange step of the path is the all-gather of the per-call lists of updated
block ids (what every rank needs to maintain the global block directory and to
schedule meshing); it runs over torch.distributed — RCCL/xGMI on GPUs ("nccl"),
gloo in the CPU tests.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

_P1, _P2, _P3 = np.uint64(73856093), np.uint64(19349663), np.uint64(83492791)


def owner_of(ids_xyz, world_size):
    """Owner rank of each block id ([k,3] int array), same function as the kernels."""
    ids = np.asarray(ids_xyz, dtype=np.int64).reshape(-1, 3).astype(np.uint64)   # sign-extended like size_t
    with np.errstate(over="ignore"):
        h = (ids[:, 0] * _P1) ^ (ids[:, 1] * _P2) ^ (ids[:, 2] * _P3)
    return (h % np.uint64(world_size)).astype(np.int64)


def allgather_block_lists(local_ids, count, cap, group=None, padded=False):
    """All-gather variable-length block-id lists.

    local_ids: int32 tensor [cap, 3] (cpu for gloo, cuda for nccl) whose first
    `count` rows are valid.  Returns a list (one per rank) of int32 tensors
    [n_r, 3] on the same device.  ONE fixed-size collective: every rank contributes its
    padded list with the count in a row of its own behind it (small messages are
    latency-bound on xGMI: a second collective for the counts would double the cost)."""
    world = dist.get_world_size(group)
    dev = local_ids.device
    if int(count) > cap or cap > local_ids.shape[0]:
        # a clamped list would silently drop block ids from every other rank's directory
        raise ValueError(f"allgather_block_lists: {count} updated blocks do not fit the {cap}-row buffer "
                         f"(size it to the map's block capacity)")
    mine = torch.empty((cap + 1, 3), dtype=torch.int32, device=dev)
    mine[:cap] = local_ids[:cap]
    mine[cap] = torch.tensor([int(count), 0, 0], dtype=torch.int32)
    flat = torch.empty((world * (cap + 1), 3), dtype=torch.int32, device=dev)   # (the concatenated form: gloo takes no other)
    dist.all_gather_into_tensor(flat, mine, group=group)
    gathered = flat.view(world, cap + 1, 3)
    counts = gathered[:, cap, 0].contiguous()
    if padded:      # the layout plvs_hip_block_directory_merge reads: [world, rows, 3] ids + [world] counts
        return gathered, counts
    c = counts.cpu()
    return [gathered[r, : int(c[r])] for r in range(world)]


def exchange_segments(send_seg, send_rec, send_run, send_counts, group=None):
    """The all-to-all of the ray-sharded integrate (TsdfChisel.shard_walk / shard_pack / shard_apply).

    send_seg [S, 8], send_rec [R, 8], send_run [U, 6] int32 tensors grouped by destination rank in rank order,
    send_counts [world, 3] (descriptors, voxel sums, runs per destination).  Returns (recv_seg, recv_rec, recv_run,
    recv_counts) grouped by source rank.  Two collectives: the counts, then ONE group of point-to-point
    operations carrying the three payloads of every pair of ranks (RCCL: a single grouped launch of ncclSend /
    ncclRecv pairs over xGMI — what an all-to-all is made of; gloo in the CPU tests); a rank's messages to itself
    are copies (at world size 1 the send buffers are handed back as they are)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = send_seg.device
    send_counts = np.ascontiguousarray(send_counts, dtype=np.int64).reshape(world, 3)
    if world == 1:      # everything stays here: the send buffers ARE the receive buffers (no copy, no count exchange)
        return send_seg.reshape(-1, 8), send_rec.reshape(-1, 8), send_run.reshape(-1, RUN_WORDS), send_counts.copy()
    sc = torch.from_numpy(send_counts.copy()).to(dev)
    rc = torch.zeros_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.cpu().numpy()
    send = [send_seg.reshape(-1, 8), send_rec.reshape(-1, 8), send_run.reshape(-1, RUN_WORDS)]
    recv = [torch.empty((int(recv_counts[:, k].sum()), w), dtype=send[k].dtype, device=dev) for k, w in enumerate((8, 8, RUN_WORDS))]
    so, ro, ops = [0, 0, 0], [0, 0, 0], []
    for p in range(world):
        for k in range(3):
            ns, nr = int(send_counts[p, k]), int(recv_counts[p, k])
            if p == rank:
                assert ns == nr
                if ns:
                    recv[k][ro[k]:ro[k] + nr].copy_(send[k][so[k]:so[k] + ns])
            else:       # both sides skip an empty message: the pairing of sends and receives stays in step
                peer = p if group is None else dist.get_global_rank(group, p)
                if ns:
                    ops.append(dist.P2POp(dist.isend, send[k][so[k]:so[k] + ns], peer, group))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, recv[k][ro[k]:ro[k] + nr], peer, group))
            so[k] += ns
            ro[k] += nr
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recv[0], recv[1], recv[2], recv_counts


RUN_WORDS = 6         # int32 words of a colour-run record on the wire (kWireRun, plvs_amd/csrc/tsdf_walk.hpp)
SAT_ROWS = 16384      # newly saturated voxels a rank reports per step (16 B each: a 256 KiB message); a longer list waits for the next step


def sharded_integrate(tsdf, d_xyz, d_rgb, d_kfid, offsets, d_Twc, group=None, timings=None):
    """One ray-sharded integrate call of this rank's TsdfChisel (every rank calls it with the same clouds).

    timings: a dict -> the call runs with a device synchronisation after every phase and ADDS each phase's wall time
    (ms) under 'walk', 'pack', 'exchange', 'apply', 'feedback' (a profiled call: slower than a plain one by the
    overlap the synchronisations remove)."""
    world = dist.get_world_size(group)
    dev = d_xyz.device
    t_last = [0.0]

    def lap(name):
        if timings is None:
            return
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        now = time.perf_counter()
        if name is not None:
            timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
        t_last[0] = now

    lap(None)
    counts = tsdf.shard_walk(d_xyz, offsets, d_Twc)
    lap("walk")
    seg = torch.empty((int(counts[:, 0].sum()), 8), dtype=torch.int32, device=dev)    # (shard_pack fills every row)
    rec = torch.empty((int(counts[:, 1].sum()), 8), dtype=torch.int32, device=dev)
    run = torch.empty((int(counts[:, 2].sum()), RUN_WORDS), dtype=torch.int32, device=dev)
    tsdf.shard_pack(seg, rec, run)
    lap("pack")
    rseg, rrec, rrun, rcounts = exchange_segments(seg, rec, run, counts, group)
    lap("exchange")
    tsdf.shard_apply(rseg, rrec, rrun, rcounts, d_rgb, d_kfid)
    lap("apply")
    # voxels whose colour saturated in this call: every rank stops sending their runs.  One fixed-size all-gather
    # (list + its length in a last row); the list is advisory — a run sent for a saturated voxel is a no-op at its
    # owner — so what does not fit waits (with the handle) for the next step.  No host read: the lengths are read on
    # the device.
    mine = torch.empty((SAT_ROWS + 1, 4), dtype=torch.int32, device=dev)       # (rows past the length are never read)
    tsdf.shard_saturated_message(mine, SAT_ROWS)
    flat = torch.empty((world * (SAT_ROWS + 1), 4), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(flat, mine, group=group)
    tsdf.shard_note_gathered(flat, world, SAT_ROWS)
    lap("feedback")
    return counts


def sharded_integrate_voxblox(tsdf, d_xyz, d_rgba, offsets, d_Twc, group=None, timings=None):
    """One ray-sharded integrate call of this rank's TsdfVoxblox ("simple"; every rank calls it with the same clouds):
    the rank casts the rays of its clouds (cloud c -> rank c % world), every voxel visit travels to the block's owner as
    a 16-byte record, the owner applies them in the reference's order (plvs_amd/csrc/tsdf_voxblox_shard.hpp).  Two
    collectives: the counts, the records.  timings: as in sharded_integrate ('walk', 'pack', 'exchange', 'apply')."""
    world = dist.get_world_size(group)
    dev = d_xyz.device
    t_last = [0.0]

    def lap(name):
        if timings is None:
            return
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        now = time.perf_counter()
        if name is not None:
            timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
        t_last[0] = now

    lap(None)
    counts = tsdf.shard_walk(d_xyz, offsets, d_Twc)
    lap("walk")
    send = torch.empty((int(counts.sum()), 4), dtype=torch.int32, device=dev)    # (shard_pack fills every row)
    tsdf.shard_pack(send)
    lap("pack")
    if world == 1:
        recv, rcounts = send, [int(counts[0])]
    else:
        recv, rcounts = _all_to_all_rows(send, counts, group)
    lap("exchange")
    tsdf.shard_apply(recv, rcounts, d_xyz, d_rgba, offsets, d_Twc)
    lap("apply")
    return counts


def _all_to_all_rows(send, send_counts, group):
    """send [sum, w] grouped by destination rank, send_counts [world] -> (recv grouped by source, recv_counts)."""
    world = dist.get_world_size(group)
    sc = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=send.device)
    rc = torch.zeros_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    rcl = [int(c) for c in rc.cpu()]
    recv = torch.zeros((sum(rcl),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send, output_split_sizes=rcl, input_split_sizes=[int(c) for c in send_counts], group=group)
    assert len(rcl) == world
    return recv, rcl


def halo_round(tsdf, missing, world, rank, exchange):
    """One round of the meshing halo: route the chunk ids this rank misses to their owners, let the owners pack the
    chunks (TsdfChisel.halo_export), bring the answers home (halo_import).  `exchange(send, send_counts) -> (recv,
    recv_counts)` is the all-to-all (torch.distributed here, tensor slices in the single-device tests)."""
    dev = getattr(tsdf, "device", None) or torch.device("cuda", torch.cuda.current_device())
    own = owner_of(missing, world) if len(missing) else np.zeros(0, np.int64)
    assert not (own == rank).any()
    order = np.argsort(own, kind="stable")
    req = torch.from_numpy(np.ascontiguousarray(missing[order], np.int32)).to(dev).reshape(-1, 3)
    req_counts = [int((own == q).sum()) for q in range(world)]
    got, got_counts = exchange(req, req_counts)                       # the ids other ranks ask this one for
    found = torch.zeros((got.shape[0],), dtype=torch.int32, device=dev)
    tsdf.halo_lookup(got.contiguous(), found)
    back_found, back_counts = exchange(found.reshape(-1, 1), got_counts)   # answers travel the reverse way, same grouping
    assert back_counts == req_counts
    back_found = back_found.reshape(-1).contiguous()
    # one payload row per chunk that exists: the owner's rows are in request order, so a requester's rows are a slice
    fh, bh = found.cpu().numpy(), back_found.cpu().numpy()
    send_rows = [int(x.sum()) for x in np.split(fh, np.cumsum(got_counts)[:-1])]
    payload = torch.empty((int(fh.sum()), tsdf.HALO_WORDS), dtype=torch.int32, device=dev)
    tsdf.halo_export(got.contiguous(), found, payload)
    back_payload, recv_rows = exchange(payload, send_rows)
    assert recv_rows == [int(x.sum()) for x in np.split(bh, np.cumsum(req_counts)[:-1])]
    tsdf.halo_import(req, back_found, back_payload.contiguous())
    return int(bh.sum())


def gather_mesh_halo(tsdf, chunk_ids, group=None, max_rounds=8):
    """Brings in the chunks of other ranks that meshing THIS rank's `chunk_ids` reads (cube faces, colour
    interpolation, gradient normals): probe, fetch what the pass reached for, probe again — until no rank misses
    anything.  Collective: every rank calls it with its own list (possibly empty).  -> chunks fetched.
    (plvs_hip_tsdf_chisel_halo_gather is the same loop behind the C ABI over an ncclComm_t.)"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = getattr(tsdf, "device", None) or torch.device("cuda", torch.cuda.current_device())
    fetched = 0
    settled = False                          # (a rank that is done keeps serving the others' requests)
    for _ in range(max_rounds):
        missing = np.zeros((0, 3), np.int32)
        if not settled:
            if tsdf.mesh_probe(chunk_ids) > 0:
                missing = tsdf.halo_missing()
            else:
                settled = True
        flag = torch.tensor([len(missing)], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            return fetched
        fetched += halo_round(tsdf, missing, world, rank, lambda send, counts: _all_to_all_rows(send, counts, group))
    raise RuntimeError("gather_mesh_halo: the halo did not settle")


def sharded_mesh_chunks(tsdf, chunk_ids, group=None):
    """ChunkManager::RecomputeMesh of THIS rank's chunks in `chunk_ids` on a sharded map: the single-device mesh of
    each chunk, byte for byte.  Every rank calls it.  -> (TsdfChisel.mesh_chunks dict, chunks fetched)."""
    fetched = gather_mesh_halo(tsdf, chunk_ids, group)
    return tsdf.mesh_chunks(chunk_ids), fetched


def voxblox_halo_ids(block_ids, world, rank):
    """The blocks of other ranks a voxblox mesh of `block_ids` reads: the seven +x / +y / +z neighbours
    (mesh_integrator.h:299-337) that this rank does not own, without duplicates.  [k, 3] int32."""
    ids = np.asarray(block_ids, dtype=np.int64).reshape(-1, 3)
    if not len(ids):
        return np.zeros((0, 3), np.int32)
    offs = np.array([[dx, dy, dz] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)][1:], np.int64)
    nb = np.unique((ids[:, None, :] + offs[None, :, :]).reshape(-1, 3), axis=0)
    return np.ascontiguousarray(nb[owner_of(nb, world) != rank], np.int32)


def sharded_mesh_blocks(tsdf, block_ids, group=None):
    """MeshIntegrator::updateMeshForBlock of THIS rank's blocks in `block_ids` on a sharded voxblox map: fetch the
    neighbour blocks the border cubes read from their owners (one round: the ids are known), then mesh.  Every rank
    calls it (with its own list, possibly empty).  -> (TsdfVoxblox.mesh_blocks dict, blocks fetched)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    need = voxblox_halo_ids(block_ids, world, rank)
    fetched = halo_round(tsdf, need, world, rank, lambda send, counts: _all_to_all_rows(send, counts, group))
    return tsdf.mesh_blocks(block_ids), fetched


class BlockDirectory:
    """plvs_block_directory (include/plvs_hip.h): the global block id -> owner rank table a rank keeps from the
    gathered lists.  Imports the HIP library on first use (this module itself needs numpy + torch only)."""

    def __init__(self, max_blocks):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self._h = ctypes.c_void_p()
        _lib.lib.plvs_hip_block_directory_create.argtypes = [ctypes.c_int, ctypes.c_void_p]
        _lib.check(_lib.lib.plvs_hip_block_directory_create(int(max_blocks), ctypes.byref(self._h)))

    def merge(self, all_ids, counts):
        ct, lib = self._ct, self._lib
        f = lib.lib.plvs_hip_block_directory_merge
        f.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p]
        lib.check(f(self._h, lib.t_ptr(all_ids), lib.t_ptr(counts), all_ids.shape[0], all_ids.shape[1],
                    lib.current_stream_ptr()))

    def count(self):
        n = self._ct.c_int()
        f = self._lib.lib.plvs_hip_block_directory_count
        f.argtypes = [self._ct.c_void_p, self._ct.c_void_p]
        self._lib.check(f(self._h, self._ct.byref(n)))
        return n.value

    def close(self):
        if self._h:
            self._lib.lib.plvs_hip_block_directory_destroy.argtypes = [self._ct.c_void_p]
            self._lib.lib.plvs_hip_block_directory_destroy(self._h)
            self._h = None
