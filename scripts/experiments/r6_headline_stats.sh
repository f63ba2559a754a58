# Round 6: kernel stats of the headline alone (20 timed steps) — bash scripts/experiments/r6_headline_stats.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-hs}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_headline.md 2>$O/summary.err
rm -rf $O/prof
head -24 $O/kernel_stats_headline.md
