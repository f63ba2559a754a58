#!/bin/bash
# Kernel time line of one single-keyframe integrate call (order-free): rocprofv3 kernel trace of walk_exp-like loop
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_b1
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench.py --batch ${1:-1} --steps 6 --warmup 4 --no-cpu-baseline --no-frontend --no-other-mode-leg --no-voxblox-leg --no-parity-check > $O/run.log 2>&1
python - <<PY
import csv
rows=[]
for r in csv.DictReader(open("$O/t_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("plvs::","").split("(")[0][:40], "k"))
try:
    for r in csv.DictReader(open("$O/t_memory_copy_trace.csv")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction","")[:20], "c"))
except Exception as e:
    print("no copy trace", e)
rows.sort()
# last call: find last walk_tiles
idx=[i for i,r in enumerate(rows) if "walk_tiles" in r[2]]
i0=idx[-1]
# back up to the previous apply/fold end
start=i0
while start>0 and rows[i0][0]-rows[start-1][1] < 60000: start-=1
t0=rows[start][0]
end=i0
while end+1 < len(rows) and rows[end+1][0]-rows[i0][0] < 400000: end+=1
for r in rows[start:end+1]:
    print(f"{(r[0]-t0)/1e3:8.1f} us  +{(r[1]-r[0])/1e3:6.1f}  {r[2]}")
PY
