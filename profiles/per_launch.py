#!/usr/bin/env python
"""Per-LAUNCH view of one kernel in a rocprofv3 kernel trace (K1's minimum launch is 3.44 us, its average 5.55 — where does the spread come from?).

  python profiles/per_launch.py <dir with *_kernel_trace.csv> <kernel name substring> [grid]

For every launch of the kernel, in time order: its duration, the idle gap on the device before it (start minus the end of the previous kernel of any name, same
queue), and what ran before it.  Then: a histogram of the durations, and the durations grouped by the preceding gap (back-to-back / behind a short gap / behind a long
idle period) — a spread that follows the gap is the state the launch finds the device in (clocks, caches, the queue's acquire), not the kernel's own chain.
"""
import csv
import glob
import os
import sys

import numpy as np


def main(d, name, grid=None):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), r.get("Queue_Id", "")))
    rows.sort()
    out = []
    prev_end, prev_name = None, ""
    for (s, e, k, g, q) in rows:
        if name in k and (grid is None or g == grid):
            out.append((e - s, (s - prev_end) if prev_end is not None else -1, prev_name[:40]))
        prev_end, prev_name = e, k
    if not out:
        print("no launch of", name)
        return
    dur = np.asarray([o[0] for o in out], float)
    gap = np.asarray([o[1] for o in out], float)
    print(f"== {d}: {len(out)} launches of *{name}*" + (f" grid={grid}" if grid else ""))
    print(f"duration ns: min {dur.min():.0f}  p10 {np.percentile(dur, 10):.0f}  p50 {np.median(dur):.0f}  mean {dur.mean():.0f}  p90 {np.percentile(dur, 90):.0f}  max {dur.max():.0f}")
    edges = [0, 3500, 4000, 4500, 5000, 5500, 6000, 7000, 8000, 10000, 1 << 40]
    h, _ = np.histogram(dur, edges)
    print("histogram (ns): " + "  ".join(f"[{edges[i]}-{edges[i + 1] if edges[i + 1] < (1 << 39) else 'inf'}) {h[i]}" for i in range(len(h)) if h[i]))
    for lo, hi, label in ((-2, 2_000, "back to back (gap < 2 us)"), (2_000, 50_000, "gap 2-50 us"), (50_000, 1_000_000, "gap 50 us - 1 ms"), (1_000_000, 1 << 60, "gap > 1 ms (idle device)")):
        m = (gap >= lo) & (gap < hi)
        if m.any():
            print(f"{label:28s} n={int(m.sum()):4d}  duration mean {dur[m].mean():7.0f}  p50 {np.median(dur[m]):7.0f}  min {dur[m].min():7.0f}  max {dur[m].max():7.0f}   (gap mean {gap[m].mean():.0f} ns)")
    print("launch#, duration_ns, gap_before_ns, kernel_before")
    for i, o in enumerate(out):
        print(f"{i},{o[0]},{o[1]},{o[2]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
