# rocprofv3 --kernel-trace --stats of the voxblox leg ALONE (bench.py --backend voxblox) -> gpurun_out/<tag>/kernel_stats_voxblox.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r05v}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--backend voxblox --batch 25 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 8 --warmup 4 $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_voxblox_under_rocprof.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_voxblox.md 2>$O/summary.err
rm -rf $O/prof
head -30 $O/kernel_stats_voxblox.md
head -c 600 $O/bench_voxblox_under_rocprof.json
