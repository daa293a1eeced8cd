#!/usr/bin/env python3
"""Condense a tools/profile_r01.sh output directory into one JSON + text summary
(per-kernel durations from the kernel trace, PMC sums per dispatch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEEP = ("k_psf_lr", "k_chief_center", "k_forward_integral", "k_trace", "k_local_psf_render", "k_sample_rays",
        "k_propagate", "k_psf_normalize")


def short(name):
    for k in KEEP:
        if k in name:
            return k
    return None


def main(d, biggest_grid_only=False):
    """biggest_grid_only: per kernel keep only the dispatches of its LARGEST grid (the workload's own launches, not the
    16-ray probes of lens set-up that go through the same kernels)."""
    out = {"kernel_trace": {}, "pmc": {}}
    for f in glob.glob(os.path.join(d, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        durs = defaultdict(list)
        meta = {}
        rows = list(csv.DictReader(open(f)))
        big = defaultdict(int)
        for r in rows:
            k = short(r["Kernel_Name"])
            if k:
                big[k] = max(big[k], int(r["Grid_Size_X"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if k and biggest_grid_only and int(r["Grid_Size_X"]) != big[k]:
                continue
            if k:
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                meta[k] = {x: r[x] for x in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                             "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
        for k, v in durs.items():
            v2 = sorted(v)
            out["kernel_trace"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": v2[0],
                                      "median_us": v2[len(v2) // 2], "max_us": v2[-1],
                                      "per_dispatch_us": [round(x, 1) for x in v], **meta[k]}
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        rows = list(csv.DictReader(open(f)))
        big = defaultdict(int)
        for r in rows:
            k = short(r["Kernel_Name"])
            if k:
                big[k] = max(big[k], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if k and biggest_grid_only and int(r["Grid_Size"]) != big[k]:
                continue
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                # steady-state dispatches are the last ones (after the speculation warm-up)
                out["pmc"].setdefault(k, {})[c] = {"dispatches": len(v), "last": v[-1],
                                                   "mean_last3": sum(v[-3:]) / len(v[-3:])}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], biggest_grid_only="--biggest-grid" in sys.argv[2:])
