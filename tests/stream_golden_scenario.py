"""The streaming headline of bench.py — 25 steps of 100 key frames of the office loop (tests/synth_scene.py), chisel 5 cm /
5 m — as a golden scenario: the REFERENCE's own open_chisel (compiled unmodified, oracle/_ref/libchisel_full_ref_o3.so) integrates
the 2 500 clouds one by one (scripts/make_stream_golden.py); after each of steps 14 .. 25 the map's exact part is digested
— sorted chunk ids, observed mask, key-frame id and colour planes of every chunk — and after steps 19 and 25 the
reference's f32 sdf / weight of 4 096 observed voxels are kept.  tests/test_stream_golden_reference.py runs the HIP depth
entry point over the same images and compares (steps 1 .. 13: tests/test_measured_configs.py)."""
import hashlib

import numpy as np

STEPS = 25
PER_STEP = 100
CHECK_FROM = 14                 # steps (1-based) whose map is digested
SAMPLE_STEPS = (19, 25)         # ... and whose sdf / weight are sampled
NSAMPLE = 4096
RES = 0.05
MAX_DEPTH = 5.0


def keyframes(step, images=False, threads=16):
    """Key frames of step `step` (1-based) of the stream."""
    from tests.synth_scene import make_stream_keyframes
    return make_stream_keyframes(PER_STEP, first=PER_STEP * (step - 1), max_depth=MAX_DEPTH, seed=0, threads=threads, images=images)


def exact_digest(chunk_ids, get_chunk):
    """sha256 over what the order-free mode reproduces exactly: chunk ids, which voxels are observed, kfid, colour."""
    ids = sorted(tuple(int(v) for v in c) for c in chunk_ids)
    h = hashlib.sha256()
    h.update(np.asarray(ids, np.int32).tobytes())
    for cid in ids:
        sdf, w, kf, col = get_chunk(*cid)
        h.update(np.ascontiguousarray(w > 0).tobytes())
        h.update(np.ascontiguousarray(kf, np.uint32).tobytes())
        h.update(np.ascontiguousarray(col, np.uint32).tobytes())
    return h.hexdigest(), len(ids)


def sample_positions(chunk_ids, get_chunk, step):
    """NSAMPLE observed voxels of the map, by a seeded draw over (sorted chunk, voxel index): ids [n, 3], voxel [n]."""
    ids = sorted(tuple(int(v) for v in c) for c in chunk_ids)
    rng = np.random.default_rng(1000 + step)
    picks_c = rng.integers(0, len(ids), 8 * NSAMPLE)
    picks_v = rng.integers(0, 4096, 8 * NSAMPLE)
    cache = {}
    out_c, out_v = [], []
    for c, v in zip(picks_c, picks_v):
        cid = ids[int(c)]
        if cid not in cache:
            cache[cid] = get_chunk(*cid)[1]
        if cache[cid][int(v)] > 0:
            out_c.append(cid)
            out_v.append(int(v))
            if len(out_c) == NSAMPLE:
                break
    return np.asarray(out_c, np.int32), np.asarray(out_v, np.int32)
