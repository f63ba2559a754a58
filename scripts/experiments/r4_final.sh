# End-of-round-4 measurements on the GPU box: the full GPU suite, the default bench line (twice), the ray-sharded step on one rank
# with its kernel stats, kernel traces of 1- and 5-key-frame calls, the batch sweep on the stream.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04f}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_$i.err | grep "^{" | tail -1 > $O/bench_$i.json; done
for i in 1 2; do timeout 300 python bench.py --sharded-at-one --steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check 2>&1 | grep "^{" | tail -1 > $O/sharded_at_one_$i.json; done
bash scripts/experiments/r4_sharded_stats.sh ${1:-r04f}_sh > $O/sharded_stats.txt 2>&1
cp gpurun_out/${1:-r04f}_sh/kernel_stats_sharded.md $O/ 2>/dev/null
bash scripts/experiments/r4_small_trace.sh ${1:-r04f}_k1 1 40 > $O/trace_1kf.txt 2>&1
bash scripts/experiments/r4_small_trace.sh ${1:-r04f}_k5 5 40 > $O/trace_5kf.txt 2>&1
[ -f scripts/experiments/r4_batch_sweep_stream.sh ] && bash scripts/experiments/r4_batch_sweep_stream.sh ${1:-r04f}_sw > $O/sweep.txt 2>&1
python - $O <<'PY'
import json, sys
o = sys.argv[1]
for i in (1, 2):
    d = json.loads(open(f"{o}/bench_{i}.json").read())
    leg = d.get("realistic_legs", {})
    print("bench", i, d["value"], d["ms_per_step"], d["roofline"]["frac"], "steady", d.get("steady_state", {}).get("value"),
          "ordered", d.get("bit_exact_mode", {}).get("ms_per_step"), "vbx", d.get("voxblox_configs3", {}).get("value"),
          "fe", d.get("frontend", {}).get("ms_per_frame"), "first_lap", leg.get("first_lap", {}).get("ms_per_call"),
          "um5", {k: v.get("ms_per_call_median") for k, v in leg.get("updatemap_5", {}).items() if isinstance(v, dict)},
          "um1", {k: v.get("ms_per_call_median") for k, v in leg.get("updatemap_1", {}).items() if isinstance(v, dict)})
    s = json.loads(open(f"{o}/sharded_at_one_{i}.json").read())
    print("sharded-at-one", i, s["value"], s["ms_per_step"], s.get("phases_ms"))
PY
