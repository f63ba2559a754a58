#!/usr/bin/env python3
"""Per-launch HBM traffic of the TSDF integrate pipeline from two rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only), written as a small JSON that
bench.py quotes in `roofline.traffic`.

    pmc_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <out.json> [voxblox]

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB-sized
units as rocprofv3 reports them (value x 1024 B); FETCH_SIZE on gfx950 counts 128-B requests
as 64 B for wide coalesced reads, so the read side is reported twice: raw and x2 (upper
bound; gathers of short pieces are closer to raw).  The pipeline total per launch = sum over
its kernels of (mean bytes per dispatch x dispatches per integrate call)."""
import csv
import json
import sys
from collections import defaultdict

PIPELINE = ("pose_prep", "walk_prologue", "walk_fast", "walk_tiles", "seg_pass", "seg_scan", "apply_chunks", "compact_runs", "sort_runs_small", "fold_colours_masks",
            "ray_count", "scan_tile_sums", "scan_sums", "scan_tile_apply", "mark_tiles", "ray_tiles",
            "radix_hist", "radix_scatter", "radix_scatter_lds", "radix_onesweep", "radix_digit_totals", "scan_single", "voxel_heads",
            "fold_colours", "reduce_sums", "run_counts", "mark_blocks", "gather_runs", "chain_runs",
            # (round 6: a long call's runs collected chunk by chunk.  NOT in the sum: the hipMemsetAsync that zeroes its run matrix,
            # 8 - 16 MB written per call — the runtime's fill kernel has one name for the bench's own fills too)
            "runs_count", "runs_rowscan", "rows_place", "runs_scatter", "parts_count", "rows_heads", "parts_place",
            "sort_runs_medium", "publish_counters")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("plvs::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0]


def load(path):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


VOXBLOX = ("vb_pose_prep", "vb_ray_pass", "vb_expand", "vb_chain_chunks", "vb_publish_counters", "vb_merge_keys", "vb_merge_bundles",
           "scan_single", "scan_tile_sums", "scan_sums", "scan_tile_apply", "radix_hist", "radix_scatter", "radix_scatter_lds",
           "radix_onesweep", "radix_digit_totals")


def main(argv):
    fetch, write = load(argv[1]), load(argv[2])
    global PIPELINE
    if len(argv) > 4 and argv[4] == "voxblox":      # the voxblox leg alone (bench.py --backend voxblox): one fold per call
        PIPELINE = VOXBLOX
        calls = fetch["vb_chain_chunks"][1] or 1
    else:
        # one ray_tiles (ordered mode) / walk_tiles (order-free mode) dispatch per integrate call
        calls = fetch["ray_tiles"][1] or fetch["walk_tiles"][1] or 1
    out = {"unit": "bytes per integrate call (one launch of the pipeline)", "integrate_calls": calls, "kernels": {}}
    tot_r = tot_w = 0.0
    for k in PIPELINE:
        if k not in fetch and k not in write:
            continue
        r = fetch[k][0] * 1024.0 / calls
        w = write[k][0] * 1024.0 / calls
        out["kernels"][k] = {"dispatches_per_call": round(fetch[k][1] / calls, 2), "fetch_raw": round(r), "fetch_x2": round(2 * r),
                             "write": round(w)}
        tot_r += r
        tot_w += w
    out["fetch_raw"] = round(tot_r)
    out["fetch_x2"] = round(2 * tot_r)
    out["write"] = round(tot_w)
    out["traffic"] = round(2 * tot_r + tot_w)
    out["note"] = "traffic = fetch_x2 + write (upper bound on the read side)"
    json.dump(out, open(argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("fetch_raw", "fetch_x2", "write", "traffic", "integrate_calls")}))


if __name__ == "__main__":
    main(sys.argv)
