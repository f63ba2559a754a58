#!/bin/bash
# A/B of an environment switch of the library on the GPU box: bash scripts/gpu_env_ab.sh <tag> <VAR> "<values>" [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; VAR="$2"; VALUES="$3"; shift 3
O="gpurun_out/$TAG"; mkdir -p "$O"
for V in $VALUES; do
  ( export $VAR=$V; timeout 300 python bench.py --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg "$@" 2>&1 | tail -3 ) > "$O/bench_${VAR}_$V.log" 2>&1
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.log")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f).read()[-600:]); continue
    r = d["roofline"]; leg = d.get("realistic_legs", {}); ss = d.get("steady_state", {})
    print(os.path.basename(f), "value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], r["stage_ms_per_launch"])
    print("   steady", ss.get("value"), ss.get("ms_per_step"), ss.get("roofline", {}).get("frac"), ss.get("stage_ms_per_launch"))
    print("   first_lap", leg.get("first_lap", {}).get("ms_per_call"), "um5", {k: v.get("ms_per_call_median") for k, v in leg.get("updatemap_5", {}).items() if isinstance(v, dict)},
          "um1", {k: v.get("ms_per_call_median") for k, v in leg.get("updatemap_1", {}).items() if isinstance(v, dict)})
PY
