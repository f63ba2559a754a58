#!/bin/bash
# A/B of developer builds of libplvs_hip.so on the GPU box: bash scripts/gpu_ab.sh <tag> "<variant names; '-' = the product>" [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; VARIANTS="$2"; shift 2
O="gpurun_out/$TAG"; mkdir -p "$O"
for V in $VARIANTS; do
  if [ "$V" = "-" ]; then unset PLVS_HIP_LIB; N=product; else export PLVS_HIP_LIB="$PWD/plvs_amd/lib/libplvs_hip_$V.so"; N=$V; fi
  ( timeout 300 python bench.py --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg "$@" 2>&1 | tail -3 ) > "$O/bench_$N.log" 2>&1
done
unset PLVS_HIP_LIB
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
