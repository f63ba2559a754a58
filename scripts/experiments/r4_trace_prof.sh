cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
PLVS_HIP_TSDF_TRACE=1 timeout 200 python bench.py --steps 6 --warmup 3 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg > $O/trace.log 2>&1
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg 2>&1 | tail -3 ) > $O/rocprof.log 2>&1
python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/kernel_stats.md 2>&1
( timeout 600 python -m pytest tests/test_tsdf_chisel.py tests/test_measured_configs.py tests/test_shard_rays.py tests/test_tsdf_golden_reference.py tests/test_tsdf_deform.py tests/test_tsdf_loadmap.py tests/test_cpp_mirror.py tests/test_tsdf_mesh.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
grep "tsdf_chisel\]" $O/trace.log | tail -8
head -40 $O/kernel_stats.md
grep -E "passed|failed|error" $O/pytest.log | tail -3
