cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4j; mkdir -p $O
( timeout 900 python -m pytest tests/test_tsdf_chisel.py tests/test_measured_configs.py tests/test_shard_rays.py tests/test_tsdf_golden_reference.py tests/test_tsdf_deform.py tests/test_tsdf_loadmap.py tests/test_cpp_mirror.py tests/test_tsdf_mesh.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -60 ) > $O/pytest.log 2>&1
grep -E "passed|failed|error|Error" $O/pytest.log | tail -5
bash scripts/gpu_ab.sh r4j "-" --steps 20 --warmup 5
