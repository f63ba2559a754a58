# A/B of the one-launch radix pass (PLVS_SORT_ONESWEEP_MIN) on the voxblox leg and the chisel headline, no profiler
A="--steps 8 --warmup 4 --no-frontend --no-cpu-baseline --no-realistic-legs --no-steady-state-leg --no-other-mode-leg --no-parity-check"
for m in 262144 0 262144 0; do
  echo "== PLVS_SORT_ONESWEEP_MIN=$m"
  PLVS_SORT_ONESWEEP_MIN=$m python bench.py --backend voxblox --batch 25 $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('voxblox ms_per_step', r['ms_per_step'])"
  PLVS_SORT_ONESWEEP_MIN=$m python bench.py --steps 20 --warmup 5 --no-voxblox-leg ${A#--steps 8 --warmup 4} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('chisel ms_per_step', r['ms_per_step'], 'gpu', r['roofline'].get('ms_per_launch'))"
done
