cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i; mkdir -p $O
( timeout 600 python -m pytest tests/test_tsdf_chisel.py tests/test_measured_configs.py tests/test_shard_rays.py tests/test_tsdf_golden_reference.py tests/test_tsdf_deform.py tests/test_tsdf_loadmap.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
bash scripts/gpu_ab.sh r4i "-" --steps 20 --warmup 5
for B in 5; do
  ( PLVS_HIP_TSDF_TRACE=1 timeout 200 python bench.py --batch $B --steps 100 --warmup 20 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -4 ) > $O/batch_$B.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4i/batch_*.log")):
    L=open(f).read().strip().splitlines()
    print(L[-2][:300])
    d=json.loads(L[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r["stage_ms_per_launch"], r["ms_per_launch"])
PY
