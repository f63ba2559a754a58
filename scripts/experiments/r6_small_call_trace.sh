# rocprofv3 kernel trace of one-key-frame calls (scripts/experiments/small_calls.py 1 2): the kernels of three late calls
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r06sc}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o r -- python $R/scripts/experiments/small_calls.py 1 2 2>&1 | tail -3 ) > $O/step.log 2>&1
python - "$(find $O/prof -name '*kernel_trace.csv' | head -1)" <<'PY' > $O/trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pro = [i for i, r in enumerate(rows) if "walk_prologue" in r["Kernel_Name"]]
for a, b in zip(pro[-5:-2], pro[-4:-1]):
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("plvs::", "").replace("void ", "")[:44]
        print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {name}")
    print()
PY
cat $O/step.log; cat $O/trace.txt
rm -rf $O/prof
