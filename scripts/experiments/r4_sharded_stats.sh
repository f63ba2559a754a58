# rocprofv3 kernel stats of the ray-sharded step on ONE rank (bench.py --sharded-at-one, the streaming workload): where the
# 3 ms of a step go, kernel by kernel.  Usage (GPU box): bash scripts/experiments/r4_sharded_stats.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r04s}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--sharded-at-one --steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_sharded.md 2>$O/summary.err
head -50 $O/kernel_stats_sharded.md
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["value"], d["ms_per_step"], d.get("phases_ms"), d["config"].get("ms_per_step_each"))
PY
rm -rf $O/prof
