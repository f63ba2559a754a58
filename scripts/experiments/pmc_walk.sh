#!/bin/bash
# PMC passes over the order-free integrate of the bench stream (walk_exp.py 0); the CSVs land in gpurun_out/pmc_walk/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_walk
mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/scripts/experiments/walk_exp.py 0 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "walk_tiles" in k or "apply_chunks" in k:
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:40s} {c:28s} {v / n:16.0f}  (n={n})")
PY
