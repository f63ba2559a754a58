#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite result (…_results.db) into the per-kernel
stats table that `--stats` would print (calls, total/avg/min/max duration)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | "
                     f"{a[3] / 1e3:.2f} | {100.0 * a[1] / total:.1f} |")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
