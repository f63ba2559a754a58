#!/bin/bash
# A/B of an environment switch incl. the other-mode and voxblox legs: bash scripts/gpu_env_ab2.sh <tag> <VAR> "<values>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; VAR="$2"; VALUES="$3"
O="gpurun_out/$TAG"; mkdir -p "$O"
for V in $VALUES; do
  ( export $VAR=$V; timeout 300 python bench.py --no-frontend --no-cpu-baseline --no-realistic-legs --no-parity-check --steps 20 --warmup 5 2>&1 | grep "^{" | tail -1 ) > "$O/bench_${VAR}_$V.json" 2>&1
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.loads(open(f).read())
    print(os.path.basename(f), "stream", d["ms_per_step"], d["roofline"]["stage_ms_per_launch"], "steady", d["steady_state"]["ms_per_step"],
          "ordered", d["bit_exact_mode"]["ms_per_step"], "voxblox", d["voxblox_configs3"]["ms_per_step"], d["voxblox_configs3"]["value"])
PY
