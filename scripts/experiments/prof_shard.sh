cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_shard -o r -- python $R/scripts/experiments/shard_rank_time.py 8 > $R/gpurun_out/prof_shard.log 2>&1
python $R/scripts/prof_summary.py $R/gpurun_out/prof_shard 2>/dev/null | head -50
