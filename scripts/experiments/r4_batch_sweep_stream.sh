cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4h; mkdir -p $O
for B in 100 25 10 5; do
  S=$((2000 / B)); W=$((500 / B))
  ( timeout 200 python bench.py --batch $B --steps $S --warmup $W --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -1 ) > $O/batch_$B.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4h/batch_*.log")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r["stage_ms_per_launch"], r["ms_per_launch"])
    except Exception as e: print(f, "FAIL", open(f).read()[-300:])
PY
