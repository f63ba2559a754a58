# rocprofv3 kernel trace of the streaming headline: the kernels of the LAST step with start offsets, durations and queue ids
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r05t}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o r -- python $R/scripts/experiments/depth_entry_stream.py 8 100 depth 2>&1 | grep "ms/step" ) > $O/step.log 2>&1
cat $O/step.log
python - "$(find $O/prof -name '*kernel_trace.csv' | head -1)" <<'PY' > $O/trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last walk_prologue on
idx = max(i for i, r in enumerate(rows) if "walk_prologue" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("plvs::", "").replace("void ", "")[:44]
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {name}")
PY
cat $O/trace.txt
rm -rf $O/prof
