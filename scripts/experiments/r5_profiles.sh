# Round 5: the evidence files of the headline (depth-image input, order-free) — run on the GPU box:
#   bash scripts/experiments/r5_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the headline ALONE (20 timed steps, no other leg) -> kernel_stats_headline.md
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of the same command -> pmc_traffic.json
#      (scripts/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, per integrate call)
# The summaries are copied to profiles/r05_* by hand; the raw traces stay on the box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r05p}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check $BENCH_ARGS"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_headline.md 2>$O/summary.err
rm -rf $O/prof
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc_$C -o r -- python $R/bench.py --steps 4 --warmup 4 $ARGS 2>&1 | tail -2 ) > $O/pmc_$C.log 2>&1
done
python scripts/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -30 $O/kernel_stats_headline.md
cat $O/pmc_traffic.log | tail -5
