#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench, rocprof stats + PMC.
# Every stage has its own timeout and log under gpurun_out/.  Usage:
#   bash scripts/gpu_check.sh [tag] [stages]     stages: subset of "test smoke bench prof pmc multi"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-run}"
STAGES="${2:-test smoke bench prof pmc}"
O="gpurun_out/$TAG"
mkdir -p "$O"
export TMPDIR=/tmp
R="$PWD"
echo "start $(date +%T)" > "$O/stages.log"
if [[ "$STAGES" == *test* ]]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v NCCL | tail -60 ) > "$O/pytest_gpu.log" 2>&1
  echo "pytest done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *smoke* ]]; then
  ( timeout 180 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20 ) > "$O/smoke.log" 2>&1
  echo "smoke done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *bench* ]]; then
  ( timeout 300 python bench.py $BENCH_ARGS 2>&1 | tail -20 ) > "$O/bench.log" 2>&1
  echo "bench done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *vbx* ]]; then
  ( timeout 300 python bench.py --backend voxblox --no-frontend 2>&1 | tail -5 ) > "$O/bench_voxblox.log" 2>&1
  echo "bench voxblox done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *sweep* ]]; then
  for B in 1 5 10 25 50 100; do
    ( timeout 200 python bench.py --batch $B --no-cpu-baseline --no-frontend 2>&1 | tail -1 ) >> "$O/sweep.log" 2>&1
  done
  echo "sweep done $(date +%T)" >> "$O/stages.log"
fi
# the first multi-GPU box: the N = 2 bench over RCCL exactly as the driver launches it (one rank per GPU), chisel ray-sharded and
# the voxblox leg; skipped where the box has one device
if [[ "$STAGES" == *multi* ]]; then
  NDEV=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
  if [ "${NDEV:-0}" -ge 2 ]; then
    ( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline --no-frontend 2>&1 | tail -3 ) > "$O/bench_2gpu.log" 2>&1
    ( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 3 --backend voxblox --no-cpu-baseline --no-frontend 2>&1 | tail -3 ) > "$O/bench_2gpu_voxblox.log" 2>&1
  else
    # one device: the same two-rank launch with both ranks on device 0 and the exchanges over gloo (bench.py, PLVS_BENCH_REHEARSAL:
    # RCCL refuses two ranks on one device) — the N = 2 code path end to end with two real processes; the times mean nothing
    for B in chisel voxblox; do
      ( PLVS_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
          --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 4 --warmup 2 --backend $B --no-cpu-baseline --no-frontend \
          2>/dev/null | tail -1 ) > "$O/bench_2ranks_one_device_$B.log" 2>&1
    done
  fi
  echo "multi done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *prof* ]]; then
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o r -- python "$R/bench.py" --no-cpu-baseline --no-frontend --no-realistic-legs $BENCH_ARGS 2>&1 | tail -5 ) > "$O/rocprof.log" 2>&1
  echo "rocprof done $(date +%T)" >> "$O/stages.log"
fi
if [[ "$STAGES" == *pmc* ]]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmc_$C" -o r -- python "$R/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-frontend --no-other-mode-leg --no-voxblox-leg --no-realistic-legs $BENCH_ARGS 2>&1 | tail -5 ) > "$O/pmc_$C.log" 2>&1
  done
  echo "pmc done $(date +%T)" >> "$O/stages.log"
fi
find "$O" -type f | head -40
cat "$O/stages.log"
for f in pytest_gpu smoke bench bench_voxblox sweep rocprof; do [ -f "$O/$f.log" ] && tail -4 "$O/$f.log"; done
if [[ "$STAGES" == *clk* ]]; then
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_clk" -o r -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-frontend 2>&1 | tail -5 ) > "$O/pmc_clk.log" 2>&1
  echo "clk done $(date +%T)" >> "$O/stages.log"
fi
