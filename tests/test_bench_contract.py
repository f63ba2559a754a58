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


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_prints_exactly_one_json_line_with_the_contract_keys():
    """The driver reads bench.py's stdout: ONE line, JSON, the contract keys + `roofline` + `cpu_baseline` — also when the
    cpu_baseline legs run the compiled reference sources, which print (open_chisel a line per garbage collection): those go to
    stderr.  A short run of the default legs' shape (two timed steps, no front end)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-frontend",
                        "--no-voxblox-leg", "--no-realistic-legs", "--no-steady-state-leg", "--no-other-mode-leg"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["unit"] == "Mvoxels/s" and d["value"] > 1000 and d["dtype"] == "f32" and "workload" in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert d.get("summary", {}).get("parity_ok") is True
