# rocprofv3 kernel trace of the headline stream (bench.py): the kernels of the last THREE steps, on a common clock
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r06ht}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 6 --warmup 6 --no-frontend --no-cpu-baseline --no-other-mode-leg --no-voxblox-leg --no-realistic-legs --no-steady-state-leg --no-parity-check 2>&1 | tail -1 | cut -c1-200 ) > $O/step.log 2>&1
python - "$(find $O/prof -name '*kernel_trace.csv' | head -1)" <<'PY' > $O/trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pro = [i for i, r in enumerate(rows) if "walk_prologue" in r["Kernel_Name"]]
for a, b in zip(pro[-4:-1], pro[-3:]):
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("plvs::", "").replace("void ", "")[:44]
        if "at::native" in name: continue
        print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {name}")
    print()
PY
cat $O/trace.txt
rm -rf $O/prof
