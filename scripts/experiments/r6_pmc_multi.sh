cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o r -- python $GRAFT_REPO_ROOT/scripts/experiments/r6_walk_prof_depth.py 8 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_sq.py /tmp/pm/sq.md $(find /tmp/pm -name "*counter_collection.csv") 2>/dev/null; grep "walk_multi\|walk_fast<2048" /tmp/pm/sq.md | cut -c1-400
