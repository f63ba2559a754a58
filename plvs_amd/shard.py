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
    [n_r, 3] on the same device.  Two fixed-size collectives: counts, then the
    padded lists (a few KB: latency-bound on xGMI, so one fused pair per call)."""
    world = dist.get_world_size(group)
    dev = local_ids.device
    if int(count) > cap or cap > local_ids.shape[0]:
        # a clamped list would silently drop block ids from every other rank's directory
        raise ValueError(f"allgather_block_lists: {count} updated blocks do not fit the {cap}-row buffer "
                         f"(size it to the map's block capacity)")
    cnt = torch.tensor([int(count)], dtype=torch.int32, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    lists = [torch.zeros_like(local_ids) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    dist.all_gather(lists, local_ids, group=group)
    if padded:      # the layout plvs_hip_block_directory_merge reads: [world, cap, 3] ids + [world] counts
        return torch.stack(lists).contiguous(), torch.cat(cnts).contiguous()
    return [lists[r][: int(cnts[r].item())] for r in range(world)]


def exchange_segments(send_seg, send_rec, send_run, send_counts, group=None):
    """The all-to-all of the ray-sharded integrate (TsdfChisel.shard_walk / shard_pack / shard_apply).

    send_seg [S, 8], send_rec [R, 8], send_run [U, 20] int32 tensors grouped by destination rank in rank order,
    send_counts [world, 3] (descriptors, voxel sums, runs per destination).  Returns (recv_seg, recv_rec, recv_run,
    recv_counts) grouped by source rank.  Four collectives: the counts, then the three payloads
    (all_to_all_single with split sizes — RCCL send/recv pairs over xGMI on GPUs, gloo in the CPU tests)."""
    world = dist.get_world_size(group)
    dev = send_seg.device
    send_counts = np.ascontiguousarray(send_counts, dtype=np.int64).reshape(world, 3)
    sc = torch.from_numpy(send_counts.copy()).to(dev)
    rc = torch.zeros_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.cpu().numpy()
    out = []
    for k, (buf, width) in enumerate(((send_seg, 8), (send_rec, 8), (send_run, 20))):
        recv = torch.zeros((int(recv_counts[:, k].sum()), width), dtype=buf.dtype, device=dev)
        dist.all_to_all_single(recv, buf.reshape(-1, width), output_split_sizes=[int(c) for c in recv_counts[:, k]],
                               input_split_sizes=[int(c) for c in send_counts[:, k]], group=group)
        out.append(recv)
    return out[0], out[1], out[2], recv_counts


def sharded_integrate(tsdf, d_xyz, d_rgb, d_kfid, offsets, d_Twc, group=None):
    """One ray-sharded integrate call of this rank's TsdfChisel (every rank calls it with the same clouds)."""
    world = dist.get_world_size(group)
    counts = tsdf.shard_walk(d_xyz, offsets, d_Twc)
    dev = d_xyz.device
    seg = torch.zeros((int(counts[:, 0].sum()), 8), dtype=torch.int32, device=dev)
    rec = torch.zeros((int(counts[:, 1].sum()), 8), dtype=torch.int32, device=dev)
    run = torch.zeros((int(counts[:, 2].sum()), 20), dtype=torch.int32, device=dev)
    tsdf.shard_pack(seg, rec, run)
    rseg, rrec, rrun, rcounts = exchange_segments(seg, rec, run, counts, group)
    tsdf.shard_apply(rseg, rrec, rrun, rcounts, d_rgb, d_kfid)
    # voxels whose colour saturated in this call: every rank stops sending their runs
    sat = tsdf.shard_saturated()
    n = torch.tensor([sat.shape[0]], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    cap = max(int(x.item()) for x in ns)
    if cap > 0:
        pad = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
        pad[: sat.shape[0]] = sat
        lists = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(lists, pad, group=group)
        for r in range(world):
            k = int(ns[r].item())
            if k:
                tsdf.shard_note_saturated(lists[r][:k].contiguous())
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
