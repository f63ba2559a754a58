# The streaming headline at other batch sizes (key frames per integrate call): 2000 key frames after 500 of warm-up each.
# Usage (GPU box): bash scripts/experiments/r4_batch_sweep_stream.sh <tag>   ->  gpurun_out/<tag>/batch_sweep_stream.jsonl
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r4h}; mkdir -p $O
: > $O/batch_sweep_stream.jsonl
for B in 1 5 10 25 50 100 250; do
  S=$((2000 / B)); W=$((500 / B))
  timeout 250 python bench.py --batch $B --steps $S --warmup $W --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg --no-parity-check 2>&1 | grep "^{" | tail -1 >> $O/batch_sweep_stream.jsonl
done
python - $O/batch_sweep_stream.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["keyframes_per_step"], d["value"], d["ms_per_step"], r["frac"], r["stage_ms_per_launch"])
PY
