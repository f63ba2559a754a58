cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4p; mkdir -p $O
for i in 1 2; do
  (timeout 300 python bench.py --steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-voxblox-leg --no-realistic-legs 2>&1 | tail -1) > $O/b$i.log
  python - $O/b$i.log <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["ms_per_step_median_max"], d["roofline"]["ms_per_launch"], d["bit_exact_mode"]["ms_per_step"], d["steady_state"]["ms_per_step"])
PY
done
timeout 300 python -m pytest tests/test_tsdf_chisel.py tests/test_shard_rays.py -m gpu -x -q 2>&1 | tail -2
