# rocprofv3 --kernel-trace --stats of the ORDERED (bit-exact) chisel mode on the stream -> gpurun_out/<tag>/kernel_stats_ordered.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r05o}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--ordered --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 8 --warmup 3 $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_ordered_under_rocprof.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_ordered.md 2>$O/summary.err
rm -rf $O/prof
head -30 $O/kernel_stats_ordered.md
head -c 500 $O/bench_ordered_under_rocprof.json
