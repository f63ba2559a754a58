#!/usr/bin/env python3
"""Issue / stall picture per kernel from rocprofv3 --pmc passes with SQ counters (counter_collection.csv files, any number of
passes): per kernel the mean per dispatch of every counter found, and the ratios that say what bounds it.

    pmc_sq.py <out.md> <counter_collection.csv> [...]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md); SQ_INSTS_* count wave
instructions; SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) = the fraction of lanes a VALU instruction keeps busy."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("plvs::", "").replace("void ", "")
    return n.split("(")[0]


def main(argv):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in argv[2:]:
        with open(path) as f:
            for row in csv.DictReader(f):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    names = sorted({c for k in acc.values() for c in k})
    with open(argv[1], "w") as out:
        out.write("| kernel | dispatches | " + " | ".join(names) + " |\n|---|---|" + "---|" * len(names) + "\n")
        rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get(names[0]))[0])
        for k, cs in rows[:24]:
            n = max(v[1] for v in cs.values())
            out.write(f"| `{k}` | {n} | " + " | ".join(f"{cs[c][0] / cs[c][1]:.4g}" if c in cs else "-" for c in names) + " |\n")
        out.write("\n| kernel | wave-instr / wave-cycle (issue) | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | VALU share of issue | LDS share | lanes per VALU instr | LDS conflict share | instr: VALU / SALU / LDS / VMEM |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, cs in rows[:24]:
            g = lambda c: cs[c][0] / cs[c][1] if c in cs and cs[c][1] else float("nan")
            wc = g("SQ_WAVE_CYCLES")
            out.write(f"| `{k}` | {g('SQ_ACTIVE_INST_ANY') / wc:.3f} | {g('SQ_WAIT_ANY') / wc:.3f} | {g('SQ_WAIT_INST_ANY') / wc:.3f} | "
                      f"{g('SQ_ACTIVE_INST_VALU') / wc:.3f} | {g('SQ_ACTIVE_INST_LDS') / wc:.3f} | "
                      f"{g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 4 * 64) * 64:.1f} | {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f} | "
                      f"{g('SQ_INSTS_VALU'):.3g} / {g('SQ_INSTS_SALU'):.3g} / {g('SQ_INSTS_LDS'):.3g} / {g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR'):.3g} |\n")


if __name__ == "__main__":
    main(sys.argv)
