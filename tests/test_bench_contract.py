"""bench.py / __graft_entry__.py keep the driver's contract (flags, defaults, no CPU fallback).  CPU only."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse(argv):
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_bench_flags_and_defaults():
    a = _parse([])
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0
    assert a.batch == 100 and a.backend == "chisel" and not a.order_free      # BASELINE's metric: bit-exact chisel
    a = _parse(["--gpus", "8", "--steps", "5", "--warmup", "3"])
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 3)


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr
    assert "{" not in r.stdout            # no result line


def test_graft_entry_points_exist():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)


def test_product_fails_loudly_without_a_device():
    """No GPU here: creating any device object must raise, not fall back."""
    if torch.cuda.is_available():
        return
    sys.path.insert(0, ROOT)
    import pytest
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfChisel
    with pytest.raises(_lib.PlvsHipError):
        TsdfChisel(0.05)
