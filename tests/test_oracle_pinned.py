"""The chisel oracle pinned by the reference's OWN sources: oracle/_ref/libchisel_ref.so is open_chisel's
Raycast.cpp, DistVoxel, ColorVoxel, QuadraticTruncator and ConstantWeighter compiled unmodified from
/root/reference against a minimal Eigen stand-in (oracle/ref/).  The restatement in oracle/tsdf_chisel.c must agree
with them bit for bit on random and adversarial inputs: traversal order, tie rules and stop test of the ray cast,
the voxel update arithmetic, the truncation and weight formulas.

The .so is built in the container that has the reference tree (oracle/ref/Makefile, __graft_entry__.build()) and
travels with the snapshot; without it the test is skipped (nothing here reads /root/reference at run time)."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libchisel_ref.so")
ORA = os.path.join(ROOT, "oracle", "liboracle.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(ORA)),
                                reason="oracle/_ref/libchisel_ref.so (built where /root/reference exists) not present")


def _libs():
    ref, ora = ctypes.CDLL(REF), ctypes.CDLL(ORA)
    f3 = ctypes.c_float * 3
    for lib, name in ((ref, "ref_chisel_raycast"), (ora, "oracle_chisel_raycast")):
        fn = getattr(lib, name)
        fn.argtypes = [f3, f3, ctypes.c_void_p, ctypes.c_int]
        fn.restype = ctypes.c_int
    for lib, pre in ((ref, "ref_chisel_"), (ora, "oracle_chisel_")):
        getattr(lib, pre + "dist_integrate").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        getattr(lib, pre + "colour_integrate_simple").argtypes = [ctypes.c_void_p] + [ctypes.c_uint8] * 4
        t = getattr(lib, pre + "truncation")
        t.argtypes, t.restype = [ctypes.c_float] * 5, ctypes.c_float
        w = getattr(lib, pre + "weight")
        w.argtypes, w.restype = [ctypes.c_float] * 3, ctypes.c_float
        dg = getattr(lib, pre + "diag")
        dg.argtypes, dg.restype = [ctypes.c_float], ctypes.c_float
    return ref, ora, f3


def _rays(rng, n):
    """Segments like the integrate loop casts (a few to a few hundred voxels long), plus the awkward ones: starts
    and ends on voxel boundaries (ties between tMax values), axis-aligned and diagonal directions, zero length."""
    start = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    length = rng.uniform(0.0, 30.0, (n, 1)).astype(np.float32)
    length[-n // 20:] = rng.uniform(100.0, 400.0, (n // 20, 1)).astype(np.float32)   # carving-length rays
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)
    end = (start + d * length).astype(np.float32)
    k = n // 8
    start[:k] = np.round(start[:k])                          # integer starts: intbound's s == floor(s) branch
    end[k:2 * k] = np.round(end[k:2 * k])
    axis = rng.integers(0, 3, 2 * k)
    for i in range(2 * k, 3 * k):                            # axis-aligned
        e = start[i].copy()
        e[axis[i - 2 * k]] += np.float32(rng.uniform(-20, 20))
        end[i] = e
    for i in range(3 * k, 4 * k):                            # exact diagonals from integer + 0.5: every tMax ties
        s = np.round(start[i]) + np.float32(0.5)
        m = np.float32(rng.integers(1, 12))
        sg = rng.choice(np.array([-1.0, 1.0], np.float32), 3)
        start[i], end[i] = s, s + sg * m
    end[4 * k:4 * k + 50] = start[4 * k:4 * k + 50]          # zero length
    start[4 * k + 50:4 * k + 100] *= np.float32(1e-3)        # tiny coordinates around the origin (negative floors)
    end[4 * k + 50:4 * k + 100] = start[4 * k + 50:4 * k + 100] + np.float32(0.25) * d[4 * k + 50:4 * k + 100]
    return start, end


def test_raycast_restatement_equals_the_reference_source():
    ref, ora, f3 = _libs()
    rng = np.random.default_rng(20260925)
    start, end = _rays(rng, 40000)
    cap = 4096
    a, b = np.zeros((cap, 3), np.int32), np.zeros((cap, 3), np.int32)
    total = longest = 0
    for s, e in zip(start, end):
        na = ref.ref_chisel_raycast(f3(*s), f3(*e), a.ctypes.data, cap)
        nb = ora.oracle_chisel_raycast(f3(*s), f3(*e), b.ctypes.data, cap)
        assert na == nb, (s, e, na, nb)
        assert na <= cap
        assert np.array_equal(a[:na], b[:nb]), (s, e)
        total += na
        longest = max(longest, na)
    assert total > 400000 and longest > 300         # the sample did exercise long walks


def test_voxel_updates_truncation_and_weight_equal_the_reference_source():
    ref, ora, _ = _libs()
    rng = np.random.default_rng(7)
    # DistVoxel::Integrate: running means as the integrate loop produces them, and raw random operands
    for trial in range(200):
        sa, wa = np.float32(99999.0), np.float32(0.0)          # DistVoxel's initial state
        sb, wb = np.float32(99999.0), np.float32(0.0)
        ra, rb = (np.array([x], np.float32) for x in (sa, sb))
        qa, qb = (np.array([x], np.float32) for x in (wa, wb))
        for _ in range(300):
            u = np.float32(rng.uniform(-0.6, 0.6))
            w = np.float32(rng.uniform(0.5, 12.0))
            ref.ref_chisel_dist_integrate(ra.ctypes.data, qa.ctypes.data, u, w)
            ora.oracle_chisel_dist_integrate(rb.ctypes.data, qb.ctypes.data, u, w)
            assert ra.view(np.uint32)[0] == rb.view(np.uint32)[0] and qa.view(np.uint32)[0] == qb.view(np.uint32)[0]
    # ColorVoxel::IntegrateSimple: every weight 0..255 against random colours, and long sequences through saturation
    for cw in range(256):
        for _ in range(40):
            p = np.array(list(rng.integers(0, 256, 3)) + [cw], np.uint8)
            q = p.copy()
            r, g, b = (int(x) for x in rng.integers(0, 256, 3))
            ref.ref_chisel_colour_integrate_simple(p.ctypes.data, r, g, b, 1)
            ora.oracle_chisel_colour_integrate_simple(q.ctypes.data, r, g, b, 1)
            assert np.array_equal(p, q), (cw, r, g, b)
    p, q = np.zeros(4, np.uint8), np.zeros(4, np.uint8)
    for _ in range(600):
        r, g, b = (int(x) for x in rng.integers(0, 256, 3))
        ref.ref_chisel_colour_integrate_simple(p.ctypes.data, r, g, b, 1)
        ora.oracle_chisel_colour_integrate_simple(q.ctypes.data, r, g, b, 1)
        assert np.array_equal(p, q)
    assert p[3] == 254                                  # frozen one short of 255, as the reference's test leaves it
    # ColorVoxel::Integrate (division + Saturate: IntegrateWorldPointCloudWithNormals' flavour) — exhaustively over
    # weight x old x new for one channel, then the three channels together; and it is NOT IntegrateSimple
    differs = 0
    for name in ("colour_integrate",):
        getattr(ref, "ref_chisel_" + name).argtypes = [ctypes.c_void_p] + [ctypes.c_uint8] * 4
        getattr(ora, "oracle_chisel_" + name).argtypes = [ctypes.c_void_p] + [ctypes.c_uint8] * 4
    for cw in range(256):
        for old in range(0, 256, 3):
            for new in range(0, 256, 5):
                p = np.array([old, 255 - old, (old * 7) & 255, cw], np.uint8)
                q, q2 = p.copy(), p.copy()
                ref.ref_chisel_colour_integrate(p.ctypes.data, new, (new * 3) & 255, 255 - new, 1)
                ora.oracle_chisel_colour_integrate(q.ctypes.data, new, (new * 3) & 255, 255 - new, 1)
                assert np.array_equal(p, q), (cw, old, new)
                ora.oracle_chisel_colour_integrate_simple(q2.ctypes.data, new, (new * 3) & 255, 255 - new, 1)
                differs += int(not np.array_equal(q, q2))
    assert differs > 0, "the two colour updates round differently somewhere (else one restatement would do)"
    # QuadraticTruncator with PLVS's constants (ChiselServer.cpp:56-59) and random ones; ConstantWeighter
    for reading in np.concatenate([np.linspace(0.0, 12.0, 5000), rng.uniform(0, 100, 5000)]).astype(np.float32):
        for q_, l_, c_, s_ in ((0.0019, -0.00152, 0.001504, 6.0), tuple(rng.uniform(-1, 1, 4))):
            x = ref.ref_chisel_truncation(q_, l_, c_, s_, reading)
            y = ora.oracle_chisel_truncation(q_, l_, c_, s_, reading)
            assert np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32)
        t = np.float32(rng.uniform(0.05, 2.0))
        assert np.float32(ref.ref_chisel_weight(1.0, 0.0, t)).view(np.uint32) == \
            np.float32(ora.oracle_chisel_weight(1.0, 0.0, t)).view(np.uint32)


def test_truncation_floor_binds_sqrt_as_the_reference_translation_unit_does():
    """Chisel.cpp:447 `2.0f * sqrt(3.0f) * resolution`: double sqrt, one rounding (static_assert in
    oracle/ref/chisel_ref_wrap.cpp); the oracle's diag must be that value for every resolution PLVS may use."""
    ref, ora, _ = _libs()
    rng = np.random.default_rng(3)
    for res in np.concatenate([np.array([0.005, 0.01, 0.015, 0.02, 0.025, 0.03, 0.04, 0.05, 0.1, 0.2], np.float32),
                               rng.uniform(0.001, 0.5, 20000).astype(np.float32)]):
        assert np.float32(ref.ref_chisel_diag(res)).view(np.uint32) == np.float32(ora.oracle_chisel_diag(res)).view(np.uint32)


def test_marching_cubes_tables_are_the_reference_sources():
    """The triangle table and the edge index pairs the oracle (and, through the committed table, the HIP mesher)
    use against the ones MarchingCubes.cpp defines."""
    ref, ora = ctypes.CDLL(REF), ctypes.CDLL(ORA)
    a, b = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
    ea, eb = np.zeros(24, np.int32), np.zeros(24, np.int32)
    ref.ref_chisel_mc_tables(a.ctypes.data_as(ctypes.c_void_p), ea.ctypes.data_as(ctypes.c_void_p))
    ora.oracle_chisel_mc_tables(b.ctypes.data_as(ctypes.c_void_p), eb.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(a, b) and np.array_equal(ea, eb)
    assert (a.reshape(256, 16)[:, -1] == -1).all() and a.max() == 11
    # the product's copy of the table (plvs_amd/csrc/mc_table.inc, compiled into tsdf_mesh.hip) holds the same numbers
    import re
    text = open(os.path.join(ROOT, "plvs_amd", "csrc", "mc_table.inc")).read()
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("//"))
    nums = np.array([int(x) for x in re.findall(r"-?\d+", text)], np.int32)
    assert np.array_equal(nums, a)


# ---------------------------------------------------------------- voxblox
VREF = os.path.join(ROOT, "oracle", "_ref", "libvoxblox_ref.so")
needs_vref = pytest.mark.skipif(not os.path.exists(VREF), reason="oracle/_ref/libvoxblox_ref.so not present")


def _vlibs():
    ref, ora = ctypes.CDLL(VREF), ctypes.CDLL(ORA)
    f3 = ctypes.c_float * 3
    for lib, pre in ((ref, "ref_voxblox_"), (ora, "oracle_voxblox_")):
        rc = getattr(lib, pre + "raycast")
        rc.argtypes = [f3, f3, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
        rc.restype = ctypes.c_int
        getattr(lib, pre + "mixed_order").argtypes = [ctypes.c_int, ctypes.c_void_p]
        bl = getattr(lib, pre + "blend")
        bl.argtypes, bl.restype = [ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32, ctypes.c_float], ctypes.c_uint32
        getattr(lib, pre + "indices").argtypes = [ctypes.c_void_p] * 4
    return ref, ora, f3


@needs_vref
def test_voxblox_raycaster_restatement_equals_the_reference_source():
    """RayCaster's constructor (ray end points for normal and clearing rays, carving on and off), setupRayCaster
    and nextRayIndex of integrator_utils.cc against vb_ray_setup / vb_ray_next of oracle/tsdf_voxblox.c."""
    ref, ora, f3 = _vlibs()
    rng = np.random.default_rng(99)
    cap = 4096
    a, b = np.zeros((cap, 3), np.int32), np.zeros((cap, 3), np.int32)
    total = 0
    for i in range(30000):
        origin = rng.uniform(-5, 5, 3).astype(np.float32)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if i % 7 == 0:                               # axis-aligned rays: two t_to_next_boundary are infinite
            d = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])
        point = (origin + d * rng.uniform(0.1, 9.0)).astype(np.float32)
        if i % 11 == 0:                              # points on voxel boundaries
            point = np.round(point * 10) / np.float32(10)
        if i % 501 == 0:
            point = origin.copy()                    # zero-length ray
        vs = float(rng.choice([0.02, 0.05, 0.1]))
        vsi = np.float32(1.0 / vs)                   # static_cast<FloatingPoint>(1.0 / voxel_size)
        clearing, carving = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        na = ref.ref_voxblox_raycast(f3(*origin), f3(*point), clearing, carving, 5.0, vsi, 0.1, a.ctypes.data, cap)
        nb = ora.oracle_voxblox_raycast(f3(*origin), f3(*point), clearing, carving, 5.0, vsi, 0.1, b.ctypes.data, cap)
        assert na == nb, (origin, point, vs, clearing, carving, na, nb)
        assert np.array_equal(a[:min(na, cap)], b[:min(nb, cap)]), (origin, point, vs, clearing, carving)
        total += na
    assert total > 500000


@needs_vref
def test_voxblox_point_order_colour_blend_and_indices_equal_the_reference_source():
    ref, ora, _ = _vlibs()
    rng = np.random.default_rng(5)
    for n in (1, 7, 1023, 1024, 1025, 2048, 5000, 76800, 307200):      # ThreadSafeIndex: groups of 1024 + a tail
        a, b = np.zeros(n, np.int64), np.zeros(n, np.int64)
        ref.ref_voxblox_mixed_order(n, a.ctypes.data)
        ora.oracle_voxblox_mixed_order(n, b.ctypes.data)
        assert np.array_equal(a, b)
        assert np.array_equal(np.sort(a), np.arange(n))
    for _ in range(200000):                                              # Color::blendTwoColors
        c1, c2 = (int(x) for x in rng.integers(0, 2 ** 32, 2, dtype=np.uint64))
        w1 = np.float32(rng.choice([0.0, rng.uniform(0, 50), rng.uniform(0, 1e4)]))
        w2 = np.float32(rng.uniform(1e-4, 100))
        assert ref.ref_voxblox_blend(c1, w1, c2, w2) == ora.oracle_voxblox_blend(c1, w1, c2, w2), (c1, w1, c2, w2)
    g = np.zeros(3, np.int32)                                            # block / local voxel index, block hash
    ba, la, bb, lb = (np.zeros(3, np.int32) for _ in range(4))
    ha, hb = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
    for _ in range(100000):
        g[:] = rng.integers(-70000, 70000, 3) if rng.random() < 0.9 else rng.integers(-40, 40, 3)
        ref.ref_voxblox_indices(g.ctypes.data, ba.ctypes.data, la.ctypes.data, ha.ctypes.data)
        ora.oracle_voxblox_indices(g.ctypes.data, bb.ctypes.data, lb.ctypes.data, hb.ctypes.data)
        assert np.array_equal(ba, bb) and np.array_equal(la, lb) and ha[0] == hb[0], g


@needs_vref
def test_voxblox_marching_cubes_equal_the_reference_source():
    """MarchingCubes::meshCube of voxblox's mesh/marching_cubes.h (vertex configuration, edge interpolation with its
    small-difference midpoint, the col+2 / col+1 / col vertex order, the flat triangle normal) and the tables of
    src/mesh/marching_cubes.cc against vb_mesh_cube and the tables of oracle/tsdf_voxblox.c — every one of the 256
    corner configurations, distances with exact zeros, near-equal pairs and values of very different size."""
    ref, ora = ctypes.CDLL(VREF), ctypes.CDLL(ORA)
    a, b = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
    ea, eb = np.zeros(24, np.int32), np.zeros(24, np.int32)
    ref.ref_voxblox_mc_tables(a.ctypes.data_as(ctypes.c_void_p), ea.ctypes.data_as(ctypes.c_void_p))
    ora.oracle_voxblox_mc_tables(b.ctypes.data_as(ctypes.c_void_p), eb.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(a, b) and np.array_equal(ea, eb)
    for lib, name in ((ref, "ref_voxblox_mesh_cube"), (ora, "oracle_voxblox_mesh_cube")):
        getattr(lib, name).argtypes = [ctypes.c_void_p] * 4
        getattr(lib, name).restype = ctypes.c_int
    rng = np.random.default_rng(17)
    offs = np.array([[0, 1, 1, 0, 0, 1, 1, 0], [0, 0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1, 1, 1]], np.float32).T   # 8 x 3
    va, na, vb, nb = (np.zeros((15, 3), np.float32) for _ in range(4))
    seen, total = set(), 0
    for it in range(40000):
        vs = np.float32(rng.choice([0.02, 0.05, 0.1]))
        base = (rng.integers(-200, 200, 3).astype(np.float32) + np.float32(0.5)) * vs
        coords = np.ascontiguousarray(base + offs * vs, np.float32)
        cfg = it % 256 if it < 2560 else int(rng.integers(0, 256))
        sign = np.array([-1.0 if (cfg >> i) & 1 else 1.0 for i in range(8)], np.float32)
        mag = rng.uniform(0.0, 0.1, 8).astype(np.float32)
        kind = it % 5
        if kind == 1:
            mag[rng.integers(0, 8, 3)] = 0.0                   # exact zeros (>= 0: outside)
        elif kind == 2:
            mag[:] = np.float32(rng.uniform(1e-8, 6e-7))       # |sdf1 - sdf2| around the 1e-6 midpoint switch
            mag += rng.uniform(0, 3e-7, 8).astype(np.float32)
        elif kind == 3:
            mag *= np.float32(10.0) ** rng.integers(-6, 1, 8).astype(np.float32)
        sdf = np.ascontiguousarray(sign * mag, np.float32)
        n1 = ref.ref_voxblox_mesh_cube(coords.ctypes.data, sdf.ctypes.data, va.ctypes.data, na.ctypes.data)
        n2 = ora.oracle_voxblox_mesh_cube(coords.ctypes.data, sdf.ctypes.data, vb.ctypes.data, nb.ctypes.data)
        assert n1 == n2 and n1 % 3 == 0, (cfg, sdf)
        assert va[:n1].tobytes() == vb[:n1].tobytes(), (cfg, sdf, coords)
        assert na[:n1].tobytes() == nb[:n1].tobytes(), (cfg, sdf, coords)
        seen.add(int(sum(1 << i for i in range(8) if sdf[i] < 0)))
        total += n1
    assert len(seen) == 256 and total > 200000


@needs_vref
def test_voxblox_bundle_order_is_the_reference_maps_iteration_order():
    """MergedTsdfIntegrator integrates its bundles in the iteration order of an AnyIndexHashMapType map.  The oracle
    (and, with the same code, the product's host side) builds a std::unordered_map with the reference's hash over a
    plain key type; here the reference's OWN container type (core/block_hash.h, compiled against the Eigen stand-in)
    is filled with the same sequences — voxel indices as a cloud produces them, with repeats, negative coordinates,
    sizes that cross many rehashes — and must iterate identically."""
    ref, ora = ctypes.CDLL(VREF), ctypes.CDLL(ORA)
    for lib, name in ((ref, "ref_voxblox_bundle_order"), (ora, "oracle_voxblox_bundle_order")):
        getattr(lib, name).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        getattr(lib, name).restype = ctypes.c_int
    rng = np.random.default_rng(23)
    for n in (1, 2, 7, 13, 14, 100, 1000, 5000, 76800, 300000):
        spread = max(3, int(round(n ** (1 / 3) * 1.5)))
        g = rng.integers(-spread, spread, (n, 3)).astype(np.int32)                 # ~ n / 4 distinct voxels, many repeats
        if n > 1000:                                                                # a surface-like sheet with large offsets
            g[:, 2] = (g[:, 0] // 3 + 40000).astype(np.int32)
        g = np.ascontiguousarray(g)
        a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
        na = ref.ref_voxblox_bundle_order(g.ctypes.data, n, a.ctypes.data)
        nb = ora.oracle_voxblox_bundle_order(g.ctypes.data, n, b.ctypes.data)
        assert na == nb == len(np.unique(g, axis=0))
        assert np.array_equal(a[:na], b[:nb]), n
        if n >= 1000:
            assert not np.array_equal(a[:na], np.sort(a[:na])), "the map does not iterate in insertion order"


# ---------------------------------------------------------------- voxblox: the integrators themselves
def _ref_map(ref, vs, carving, method):
    ref.ref_voxblox_create.restype = ctypes.c_void_p
    ref.ref_voxblox_create.argtypes = [ctypes.c_float] * 5 + [ctypes.c_int, ctypes.c_char_p]
    return ctypes.c_void_p(ref.ref_voxblox_create(vs, 0.1, 10000.0, 0.1, 5.0, int(carving), method.encode()))


def _ref_blocks(ref, h):
    ref.ref_voxblox_num_blocks.argtypes = [ctypes.c_void_p]
    ref.ref_voxblox_block_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ref.ref_voxblox_get_block.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    n = ref.ref_voxblox_num_blocks(h)
    ids = np.zeros((max(n, 1), 3), np.int32)
    ref.ref_voxblox_block_ids(h, ids.ctypes.data)
    out = {}
    for b in ids[:n]:
        d, w, c = np.zeros(4096, np.float32), np.zeros(4096, np.float32), np.zeros(4096, np.uint32)
        assert ref.ref_voxblox_get_block(h, int(b[0]), int(b[1]), int(b[2]), d.ctypes.data, w.ctypes.data, c.ctypes.data)
        out[tuple(int(v) for v in b)] = (d, w, c)
    return out


@needs_vref
@pytest.mark.parametrize("method,vs,carving,far", [("simple", 0.05, False, False), ("simple", 0.10, True, True),
                                                    ("merged", 0.05, False, False), ("merged", 0.10, True, True)])
def test_voxblox_integrators_equal_the_reference_sources(method, vs, carving, far):
    """Whole clouds through voxblox's OWN SimpleTsdfIntegrator / MergedTsdfIntegrator (tsdf_integrator.cc with Layer /
    Block, compiled unmodified against the stand-ins) and through the restatement: every voxel of every block bit for
    bit — updateTsdfVoxel, computeDistance, getVoxelWeight with the drop-off, isPointValid, clearing rays, the mixed
    visiting order, the merged bundles and their hash-map order, block allocation."""
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes
    ref = ctypes.CDLL(VREF)
    oracle = oracle_lib.load()
    ref.ref_voxblox_integrate.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
    ref.ref_voxblox_destroy.argtypes = [ctypes.c_void_p]
    oracle.lib.oracle_voxblox_pose_quat.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    h = _ref_map(ref, vs, carving, method)
    ora = oracle.voxblox(vs, carving=carving)
    kfs = make_keyframes(3, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=11) if far else make_keyframes(3, seed=11)
    for k in kfs:
        xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
        rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::2], np.full((len(xyz), 1), 255, np.uint8)], 1))
        Twc = np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)
        q = np.zeros(4, np.float32)
        oracle.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
        t = np.ascontiguousarray(Twc[:, 3])
        ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))
        (ora.integrate_merged if method == "merged" else ora.integrate)(xyz, rgba, Twc)
    want = _ref_blocks(ref, h)
    got = {tuple(int(v) for v in b): ora.get_chunk(*b) for b in ora.chunk_ids()}
    assert set(got) == set(want) and len(want) > 20
    for bid, planes in want.items():
        for name, x, y in zip(("distance", "weight", "colour"), planes, got[bid]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
    ref.ref_voxblox_destroy(h)


@needs_vref
@pytest.mark.parametrize("vs,carving,far,stride", [(0.05, False, False, 2), (0.10, True, True, 2), (0.05, True, False, 3),
                                                   (0.02, False, False, 1)])
def test_voxblox_fast_integrator_equals_the_reference_source(vs, carving, far, stride):
    """FastTsdfIntegrator (tsdf_integrator.cc:505-605) — PLVS's YAML default — compiled unmodified, integrator_threads = 1,
    against the restatement with the reference's own approximate sets: the start-voxel test at half the voxel size, the
    cast from the surface end, the stop at the third already-seen voxel in a row, ApproxHashSet's slot / offset rules
    over several scans — every voxel of every block bit for bit.  And the class statement the device's integrator rests
    on: the same algorithm with collision-free sets differs from it only where two indices shared a slot."""
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes
    ref = ctypes.CDLL(VREF)
    oracle = oracle_lib.load()
    ref.ref_voxblox_integrate.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
    ref.ref_voxblox_destroy.argtypes = [ctypes.c_void_p]
    oracle.lib.oracle_voxblox_pose_quat.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    h = _ref_map(ref, vs, carving, "fast")
    ora, exact, simple = oracle.voxblox(vs, carving=carving), oracle.voxblox(vs, carving=carving), oracle.voxblox(vs, carving=carving)
    kfs = make_keyframes(4, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=11) if far else make_keyframes(4, seed=11)
    for k in kfs:
        xyz = np.ascontiguousarray(k["xyz"][::stride], np.float32)
        rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::stride], np.full((len(xyz), 1), 255, np.uint8)], 1))
        Twc = np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)
        q = np.zeros(4, np.float32)
        oracle.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
        t = np.ascontiguousarray(Twc[:, 3])
        ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))
        ora.integrate_fast(xyz, rgba, Twc, approx_sets=True)
        exact.integrate_fast(xyz, rgba, Twc, approx_sets=False)
        simple.integrate(xyz, rgba, Twc)
    want = _ref_blocks(ref, h)
    got = {tuple(int(v) for v in b): ora.get_chunk(*b) for b in ora.chunk_ids()}
    assert set(got) == set(want) and len(want) > 20
    for bid, planes in want.items():
        for name, x, y in zip(("distance", "weight", "colour"), planes, got[bid]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
    # fast really is another integrator: fewer updates than simple ...
    assert ora.last_visits() < 0.8 * simple.last_visits()
    # ... and what the sets' collisions cost: with collision-free sets (same algorithm) the map observes nearly the same
    # voxels, at distances millimetres away (the device reproduces the approximate sets, not these)
    ex = {tuple(int(v) for v in b): exact.get_chunk(*b) for b in exact.chunk_ids()}
    both = only = 0
    dd = []
    for bid, (d, w, c) in got.items():
        e = ex.get(bid)
        we = e[1] if e is not None else np.zeros(4096, np.float32)
        m = (w > 0) & (we > 0)
        both += int(m.sum())
        only += int(((w > 0) ^ (we > 0)).sum())
        if e is not None:
            dd.append(np.abs(d[m] - e[0][m]))
    dd = np.concatenate(dd)
    assert only < 0.05 * both and np.percentile(dd, 90) < 0.004 and ((dd > 0).any() or only > 0), (both, only, np.percentile(dd, 90))
    ref.ref_voxblox_destroy(h)


@needs_vref
def test_voxblox_world_cloud_integrate_equals_the_reference_source():
    """TsdfIntegratorBase::integrateWorlPointCloud (the LoadMap path) of the reference against the restatement: posed
    clouds with un-normalised and zero normals, onto a map that already holds camera-ray integrations."""
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes
    from tests.test_tsdf_loadmap import surface_cloud
    ref = ctypes.CDLL(VREF)
    oracle = oracle_lib.load()
    ref.ref_voxblox_integrate.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
    ref.ref_voxblox_integrate_world.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int]
    ref.ref_voxblox_destroy.argtypes = [ctypes.c_void_p]
    oracle.lib.oracle_voxblox_pose_quat.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    vs = 0.05
    h = _ref_map(ref, vs, False, "simple")
    ora = oracle.voxblox(vs)

    def pose(Twc):
        Twc = np.ascontiguousarray(Twc, np.float32).reshape(3, 4)
        q = np.zeros(4, np.float32)
        oracle.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
        return Twc, q, np.ascontiguousarray(Twc[:, 3])

    k = make_keyframes(1, seed=4)[0]
    xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
    rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::2], np.full((len(xyz), 1), 255, np.uint8)], 1))
    Twc, q, t = pose(k["Twc"])
    ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))
    ora.integrate(xyz, rgba, Twc)
    T2 = np.array([[0.0, -1.0, 0.0, 0.3], [1.0, 0.0, 0.0, -0.2], [0.0, 0.0, 1.0, 0.1]], np.float32)
    for seed, P in ((1, np.eye(4, dtype=np.float32)[:3]), (2, T2)):
        xyz, rgb, _, nrm = surface_cloud(20000, seed, vs)
        rgba = np.ascontiguousarray(np.concatenate([rgb, np.full((len(rgb), 1), 200, np.uint8)], 1))
        Twc, q, t = pose(P)
        ref.ref_voxblox_integrate_world(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, nrm.ctypes.data, len(xyz))
        ora.integrate_world_normals(xyz, rgba, nrm, Twc)
    # integrateWorlPointCloud leaves the blocks it creates in the integrator's temp_block_map_ — it never calls
    # updateLayerWithStoredBlocks (tsdf_integrator.cc:35-82 vs :288) — so they join the layer with the NEXT
    # integratePointCloud call; an empty cloud does it.  (The restatement and the device path insert them at once: §6.)
    before_flush = len(_ref_blocks(ref, h))
    ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, 0)
    want = _ref_blocks(ref, h)
    assert before_flush < len(want)
    got = {tuple(int(v) for v in b): ora.get_chunk(*b) for b in ora.chunk_ids()}
    assert set(got) == set(want) and len(want) > 50
    for bid, planes in want.items():
        for name, x, y in zip(("distance", "weight", "colour"), planes, got[bid]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
    ref.ref_voxblox_destroy(h)


@needs_vref
def test_voxblox_world_cloud_block_visibility_equals_the_reference_source():
    """With set_deferred_world_blocks the restatement shows what the reference's layer shows at every point of
    camera cloud -> world cloud -> world cloud -> camera cloud: the blocks a world cloud creates are absent from the
    block list (and read as missing) until the next integratePointCloud, their voxels accumulating meanwhile."""
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes
    from tests.test_tsdf_loadmap import surface_cloud
    ref = ctypes.CDLL(VREF)
    oracle = oracle_lib.load()
    ref.ref_voxblox_integrate.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
    ref.ref_voxblox_integrate_world.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int]
    ref.ref_voxblox_destroy.argtypes = [ctypes.c_void_p]
    oracle.lib.oracle_voxblox_pose_quat.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    vs = 0.05
    h = _ref_map(ref, vs, False, "simple")
    ora = oracle.voxblox(vs)
    ora.set_deferred_world_blocks(True)

    def same():
        want = _ref_blocks(ref, h)
        got = {tuple(int(v) for v in b): ora.get_chunk(*b) for b in ora.chunk_ids()}
        assert set(got) == set(want) and ora.num_chunks() == len(want)
        for bid, planes in want.items():
            for name, x, y in zip(("distance", "weight", "colour"), planes, got[bid]):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
        return len(want)

    def pose(Twc):
        Twc = np.ascontiguousarray(Twc, np.float32).reshape(3, 4)
        q = np.zeros(4, np.float32)
        oracle.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
        return Twc, q, np.ascontiguousarray(Twc[:, 3])

    kfs = make_keyframes(2, seed=4)
    counts = []
    for step in ("cam0", "world1", "world2", "cam1"):
        if step.startswith("cam"):
            k = kfs[int(step[3])]
            xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
            rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::2], np.full((len(xyz), 1), 255, np.uint8)], 1))
            Twc, q, t = pose(k["Twc"])
            ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))
            ora.integrate(xyz, rgba, Twc)
        else:
            xyz, rgb, _, nrm = surface_cloud(20000, int(step[5]), vs)
            rgba = np.ascontiguousarray(np.concatenate([rgb, np.full((len(rgb), 1), 200, np.uint8)], 1))
            Twc, q, t = pose(np.eye(4, dtype=np.float32)[:3])
            ref.ref_voxblox_integrate_world(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, nrm.ctypes.data, len(xyz))
            ora.integrate_world_normals(xyz, rgba, nrm, Twc)
        counts.append(same())
    assert counts[1] == counts[0] or counts[2] <= counts[3]        # (world clouds add nothing visible ...)
    assert counts[3] > counts[2]                                    # (... until the camera cloud publishes their blocks)
    ref.ref_voxblox_destroy(h)


@needs_vref
def test_voxblox_mesh_integrator_equals_the_reference_source():
    """MeshIntegrator<TsdfVoxel>::updateMeshForBlock of the reference (mesh/mesh_integrator.h with Layer / Block / Mesh,
    compiled unmodified) on a map its own SimpleTsdfIntegrator built, against oracle_voxblox_mesh_block on the oracle's
    copy of that map: the walk over the block, the corner gathering through neighbour blocks, unobserved corners,
    meshCube, the colour look-up — vertices, normals and colours of every block, byte for byte."""
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes
    from tests.test_tsdf_voxblox_mesh import mesh_block
    ref = ctypes.CDLL(VREF)
    oracle = oracle_lib.load()
    ref.ref_voxblox_integrate.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
    ref.ref_voxblox_mesh_block.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int]
    ref.ref_voxblox_destroy.argtypes = [ctypes.c_void_p]
    oracle.lib.oracle_voxblox_pose_quat.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    total = 0
    for vs in (0.05, 0.10):
        h = _ref_map(ref, vs, False, "simple")
        ora = oracle.voxblox(vs)
        for k in make_keyframes(3, seed=13):
            xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
            rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::2], np.full((len(xyz), 1), 255, np.uint8)], 1))
            Twc = np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)
            q = np.zeros(4, np.float32)
            oracle.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
            t = np.ascontiguousarray(Twc[:, 3])
            ref.ref_voxblox_integrate(h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))
            ora.integrate(xyz, rgba, Twc)
        cap = 4096 * 15
        v, n = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
        c = np.zeros((cap, 4), np.uint8)
        for bid in ora.chunk_ids():
            nv = ref.ref_voxblox_mesh_block(h, int(bid[0]), int(bid[1]), int(bid[2]), v.ctypes.data, n.ctypes.data, c.ctypes.data, cap)
            ov, on, oc = mesh_block(ora, *bid)
            assert nv == len(ov), (vs, bid, nv, len(ov))
            assert v[:nv].tobytes() == ov.tobytes() and n[:nv].tobytes() == on.tobytes() and c[:nv].tobytes() == oc.tobytes(), (vs, bid)
            total += nv
        assert ref.ref_voxblox_mesh_block(h, 999, 999, 999, v.ctypes.data, n.ctypes.data, c.ctypes.data, cap) == 0
        ref.ref_voxblox_destroy(h)
    assert total > 15000


# ------------------------------------------------------------------ LBD matcher (multi-index hashing k-NN)
LREF = os.path.join(ROOT, "oracle", "_ref", "liblbd_matcher_ref.so")
needs_lref = pytest.mark.skipif(not os.path.exists(LREF), reason="oracle/_ref/liblbd_matcher_ref.so not present")


def _codes(rng, n, clustered):
    """LBD-like 256-bit codes: random, or few cluster centres with a few flipped bits (many equidistant neighbours:
    the case where WHICH of them the hash search reports is decided by its probing order)."""
    if not clustered:
        return rng.integers(0, 256, (n, 32), dtype=np.uint8)
    centres = rng.integers(0, 256, (max(n // 12, 1), 32), dtype=np.uint8)
    out = centres[rng.integers(0, len(centres), n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, rng.integers(0, 4)):
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


@needs_lref
@pytest.mark.parametrize("nq,nt,seed,clustered", [(100, 100, 1, False), (300, 200, 2, True), (64, 1000, 3, True),
                                                  (500, 37, 4, False), (7, 2, 5, False), (200, 300, 6, True)])
def test_lbd_mih_knn_restatement_equals_the_reference_source(nq, nt, seed, clustered):
    """BinaryDescriptorMatcher::knnMatch(k = 2) -> Mihasher populate / batchquery / query of the reference, compiled
    unmodified, against oracle_knn2_mih: indices (incl. the choice among equidistant train lines, duplicates and exact
    copies of the query) and distances of both neighbours of every query, with and without a query mask."""
    from tests import oracle_lib
    ref = ctypes.CDLL(LREF)
    ref.ref_lbd_knn2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    oracle = oracle_lib.load()
    rng = np.random.default_rng(seed)
    t = _codes(rng, nt, clustered)
    q = _codes(rng, nq, clustered)
    q[::5] = t[rng.integers(0, nt, len(q[::5]))]            # exact copies: distance 0, often shared by several train rows
    for mask in (None, (rng.random(nq) < 0.7).astype(np.uint8)):
        idx = np.full((nq, 2), -7, np.int32)
        dist = np.full((nq, 2), -7, np.int32)
        ref.ref_lbd_knn2(q.ctypes.data, nq, t.ctypes.data, nt, None if mask is None else mask.ctypes.data, idx.ctypes.data,
                         dist.ctypes.data)
        oidx, odist = oracle.knn2(q, t, mask, mih=True)
        assert np.array_equal(idx, oidx), np.argwhere(idx != oidx)[:5]
        assert np.array_equal(dist, odist)
        assert (idx[mask.astype(bool) if mask is not None else slice(None)] >= 0).all()
