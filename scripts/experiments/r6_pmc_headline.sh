# Round 6: FETCH_SIZE / WRITE_SIZE passes of the headline alone -> pmc_traffic_chisel_order_free.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r06pmc}"; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc_$C -o r -- python $R/bench.py --steps 4 --warmup 4 $ARGS 2>&1 | tail -2 ) > $O/pmc_$C.log 2>&1
done
python scripts/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_chisel_order_free.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -3 $O/pmc_traffic.log
