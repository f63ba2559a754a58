# rocprofv3 kernel stats of the HEADLINE alone (the streaming order-free steps, no other leg): per-kernel averages that can be
# set against the line's stage_ms_per_launch.  Usage (GPU box): bash scripts/experiments/r4_headline_stats.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r04h}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_headline.md 2>$O/summary.err
head -24 $O/kernel_stats_headline.md
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms_per_launch"])
PY
rm -rf $O/prof
