# rocprofv3 kernel stats of the voxblox streaming leg under two builds of the library (plvs_amd/lib/libplvs_hip_old.so = the
# build to compare with): which kernel a change slowed down.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
for V in old new; do
  O=gpurun_out/vbs_$V; mkdir -p $O
  if [ "$V" = "old" ]; then export PLVS_HIP_LIB="$R/plvs_amd/lib/libplvs_hip_old.so"; else unset PLVS_HIP_LIB; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --backend voxblox --resolution 0.02 --batch 25 --steps 8 --warmup 4 --max-depth 8 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-voxblox-leg --no-parity-check > /dev/null 2>&1 )
  python scripts/prof_summary.py $(find $O/prof -name "*kernel_stats.csv" | head -1) > $O/stats.md 2>/dev/null
  rm -rf $O/prof
  echo "== $V"; head -22 $O/stats.md
done
