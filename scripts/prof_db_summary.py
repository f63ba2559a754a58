#!/usr/bin/env python3
"""Per-kernel calls / average / total GPU time from a rocprofv3 results .db (rocpd schema):
   prof_db_summary.py <results.db> [skip_first_n_dispatches_per_kernel]"""
import sqlite3
import sys
from collections import defaultdict


def main(argv):
    db = sqlite3.connect(argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    names = {r[0]: r[1] for r in cur.execute(f"select id, {name_col} from {sym}")}
    acc = defaultdict(list)
    for kid, st, en in cur.execute(f"select kernel_id, start, end from {disp} order by start"):
        n = names.get(kid, str(kid)).replace("(anonymous namespace)::", "").replace("plvs::", "").split("(")[0]
        acc[n].append((en - st) / 1e3)
    skip = int(argv[2]) if len(argv) > 2 else 0
    rows = []
    for n, v in acc.items():
        v = v[skip:] if len(v) > skip else v
        rows.append((sum(v), n, len(v), sum(v) / len(v), min(v), max(v)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"| kernel | calls | avg us | min us | max us | total us | % |\n|---|---|---|---|---|---|---|")
    for t, n, c, a, mn, mx in rows:
        print(f"| {n[:80]} | {c} | {a:.1f} | {mn:.1f} | {mx:.1f} | {t:.0f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv)
