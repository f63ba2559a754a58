# A/B of the voxblox streaming leg under two builds of the library: put the build to compare with at plvs_amd/lib/libplvs_hip_old.so
# (e.g. git archive <commit> plvs_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/plvs_amd/csrc), then run on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for V in old new old new; do
  if [ "$V" = "old" ]; then export PLVS_HIP_LIB="$PWD/plvs_amd/lib/libplvs_hip_old.so"; else unset PLVS_HIP_LIB; fi
  timeout 300 python bench.py --backend voxblox --resolution 0.02 --batch 25 --steps 8 --warmup 4 --max-depth 8 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'])"
done
