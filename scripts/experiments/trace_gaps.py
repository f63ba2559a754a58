#!/usr/bin/env python3
"""rocprofv3 kernel trace -> the kernels of the LAST N launches' window: start offset, duration, gap before (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    busy += e - s
    prev_end = e
print(f"window {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us")
