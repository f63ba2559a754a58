# SQ issue / stall counters of the streaming headline's kernels.  Usage (GPU box): bash scripts/experiments/r4_sq.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r04sq}"; shift; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
ARGS="--steps 4 --warmup 2 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg $*"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$O/sq$i -o r -- python $R/bench.py $ARGS 2>&1 | tail -3 ) > $O/sq$i.log 2>&1
done
python scripts/pmc_sq.py $O/sq.md $(find $O -name "*counter_collection.csv") 2> $O/sq.err
cat $O/sq.md | cut -c1-400 | tail -30; cat $O/sq.err | tail -5
rm -rf $O/sq1 $O/sq2 $O/sq3
