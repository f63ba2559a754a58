cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -3 ) > $O/rocprof.log 2>&1
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats.md 2>&1
head -14 $O/kernel_stats.md
