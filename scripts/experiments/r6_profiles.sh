# Round 6: the evidence files — run on the GPU box: bash scripts/experiments/r6_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the headline ALONE (20 timed steps)            -> kernel_stats_headline.md
#   2. FETCH_SIZE / WRITE_SIZE passes of the same command (separate runs, kernel trace)   -> pmc_traffic_chisel_order_free.json
#   3. the voxblox leg alone: kernel stats and its two PMC passes                          -> kernel_stats_voxblox.md, pmc_traffic_voxblox.json
# (scripts/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, per integrate call).  Summaries are copied to profiles/r06_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r06p}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 $ARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_headline.md 2>$O/summary.err
rm -rf $O/prof
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc_$C -o r -- python $R/bench.py --steps 4 --warmup 4 $ARGS 2>&1 | tail -2 ) > $O/pmc_$C.log 2>&1
done
python scripts/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_chisel_order_free.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
VARGS="--backend voxblox --batch 25 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-parity-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/vprof -o r -- python $R/bench.py --steps 8 --warmup 4 $VARGS 2>&1 | grep "^{" | tail -1 ) > $O/bench_voxblox_under_rocprof.json 2>> $O/rocprof.err
python scripts/prof_summary.py $(find $O/vprof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_voxblox.md 2>>$O/summary.err
rm -rf $O/vprof
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/vpmc_$C -o r -- python $R/bench.py --steps 4 --warmup 4 $VARGS 2>&1 | tail -2 ) > $O/vpmc_$C.log 2>&1
done
python scripts/pmc_traffic.py $(find $O/vpmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/vpmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_voxblox.json voxblox > $O/pmc_traffic_voxblox.log 2>&1
rm -rf $O/vpmc_FETCH_SIZE $O/vpmc_WRITE_SIZE
head -14 $O/kernel_stats_headline.md; tail -3 $O/pmc_traffic.log; head -10 $O/kernel_stats_voxblox.md; tail -3 $O/pmc_traffic_voxblox.log
