#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace / PMC counter collection) into the per-kernel table committed under profiles/.

  python profiles/summarize.py <dir with *_kernel_trace.csv [and *_counter_collection.csv]> [...]

Per kernel and grid size (the same kernel is launched with different grids, e.g. K1 with and without the K2 ride-along
workgroups): launches, average duration (ns) from the kernel trace, and the average of every collected counter.
FETCH_SIZE / WRITE_SIZE are in KiB per launch as rocprofv3 reports them (MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE
counts a wide coalesced read stream at half its bytes — the `x2` column applies that correction; other widths are
uncalibrated, so treat the corrected figure as an upper bound for 8-byte-per-lane streams).
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def grid_of(r) -> int:
    """total work-items of the launch (kernel trace: Grid_Size_X/Y/Z; counter collection: Grid_Size)"""
    if "Grid_Size" in r:
        return int(r["Grid_Size"])
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])


def main(dirs):
    for d in dirs:
        print(f"== {d}")
        dur = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"]) + f" grid={grid_of(r)}"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        ctr = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                ctr[short(r["Kernel_Name"]) + f" grid={grid_of(r)}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not dur:  # a counter-only pass (--pmc without a trace domain): launches and durations from the counter rows themselves
            seen = set()
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    key = (f, r.get("Dispatch_Id"))
                    if key in seen:
                        continue
                    seen.add(key)
                    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if r.get("End_Timestamp") and r.get("Start_Timestamp") else 0
                    dur[short(r["Kernel_Name"]) + f" grid={grid_of(r)}"].append(t)
        names = sorted(dur, key=lambda k: -sum(dur[k]))
        cols = sorted({c for k in ctr for c in ctr[k]})
        hdr = ["kernel", "launches", "avg_ns", "min_ns", "max_ns"] + [f"{c}_avg" for c in cols] + (["FETCH_SIZE_x2_bytes"] if "FETCH_SIZE" in cols else []) + (["WRITE_SIZE_bytes"] if "WRITE_SIZE" in cols else [])
        print(",".join(hdr))
        for k in names:
            v = dur[k]
            row = [k, str(len(v)), f"{sum(v) / len(v):.0f}", str(min(v)), str(max(v))]
            for c in cols:
                vals = ctr[k].get(c, [])
                row.append(f"{sum(vals) / len(vals):.2f}" if vals else "")
            if "FETCH_SIZE" in cols:
                vals = ctr[k].get("FETCH_SIZE", [])
                row.append(f"{2 * 1024 * sum(vals) / len(vals):.0f}" if vals else "")
            if "WRITE_SIZE" in cols:
                vals = ctr[k].get("WRITE_SIZE", [])
                row.append(f"{1024 * sum(vals) / len(vals):.0f}" if vals else "")
            print(",".join(row))


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
