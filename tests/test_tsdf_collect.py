"""The collected colour chain of a long order-free call (tsdf_walk.hpp: runs_count ... parts_place) against the
sorted chain of rounds 4-5: the same calls leave the same map, bit for bit, after every call.

PLVS_TSDF_COLLECT is a developer switch read once per process (1: long calls only, the default; 0: never; 2: every call that
can) — the scenario (tests/collect_scenario.py) runs in a process per setting.  With 2 the chain takes calls of 150 to 1 200
tiles on its own counts (the handle's first call), queued without them (the later ones), and must leave the calls whose tiles
reach walk_tiles (2 cm voxels: a tile of a far surface overflows every lean table) to the general chain."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, **extra):
    env = dict(os.environ, PLVS_TSDF_COLLECT=str(mode), PLVS_HIP_TSDF_TRACE="1", **extra)
    p = subprocess.run([sys.executable, "-m", "tests.collect_scenario"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln.split() for ln in p.stdout.strip().splitlines() if len(ln.split()) == 3]
    chains = [ln.split("chain ")[1].split()[0:2] for ln in p.stderr.splitlines() if "[tsdf_chisel]" in ln and "chain " in ln]
    return lines, chains


@pytest.mark.gpu
def test_hip_collected_chain_leaves_the_map_of_the_sorted_chain():
    forced, forced_chains = _run(2)
    plain, plain_chains = _run(0)
    assert len(forced) == len(plain) == 9
    for i, (a, b) in enumerate(zip(forced, plain)):
        assert a == b, f"call {i}: collected {a} vs sorted {b}"
        assert int(a[0]) > 0 and int(a[1]) > 0
    # the path under test ran: collected (2) for the depth calls, the general chain for the calls of several clouds
    kinds = [c[0] for c in forced_chains]
    assert kinds.count("2") >= 3, forced_chains
    assert kinds.count("0") >= 1, forced_chains   # (a call with tiles left to walk_tiles: handed over)
    assert all(c[0] != "2" for c in plain_chains), plain_chains


@pytest.mark.gpu
def test_hip_collected_chain_repeats_as_the_general_one_when_its_matrix_is_too_small():
    """Rows for eight chunks only (PLVS_TSDF_COLLECT_MAX_ROWS): every call that updates more sets `skip` on the device, its
    fold leaves at once, and the host runs the general chain once the call's counters are read — the same maps."""
    small, small_chains = _run(2, PLVS_TSDF_COLLECT_MAX_ROWS="8")
    plain, _ = _run(0)
    assert small == plain and len(small) == 9
    assert any(len(c) > 1 and c[1] == "REPEATED" for c in small_chains), small_chains
