# rocprofv3 evidence of the round: kernel stats of the default bench, PMC traffic of the streaming headline (order-free),
# the ordered mode and the voxblox leg.  Usage (on the GPU box): bash scripts/experiments/r4_profiles.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r04}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
COMMON="--steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-realistic-legs"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py $COMMON 2>&1 | tail -2 ) > $O/rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc_$C -o r -- python $R/bench.py --steps 4 --warmup 2 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -2 ) > $O/pmc_$C.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmco_$C -o r -- python $R/bench.py --ordered --steps 4 --warmup 2 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -2 ) > $O/pmco_$C.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmcv_$C -o r -- python $R/bench.py --backend voxblox --batch 25 --steps 4 --warmup 2 --no-frontend --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmcv_$C.log 2>&1
done
f() { find $O/$1 -name "*counter_collection.csv" | head -1; }
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) > $O/kernel_stats_pmc.md 2>$O/summary.err
python scripts/pmc_traffic.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_traffic_chisel_order_free.json
python scripts/pmc_traffic.py $(f pmco_FETCH_SIZE) $(f pmco_WRITE_SIZE) $O/pmc_traffic_chisel_ordered.json
python scripts/pmc_traffic.py $(f pmcv_FETCH_SIZE) $(f pmcv_WRITE_SIZE) $O/pmc_traffic_voxblox.json voxblox
head -30 $O/kernel_stats_pmc.md
rm -rf $O/pmc_*/ $O/pmco_*/ $O/pmcv_*/   # (the raw counter CSVs are large)
