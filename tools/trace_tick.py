#!/usr/bin/env python
"""Print the GPU-side timeline (kernels + memory copies) of the last tick in a rocprofv3 --kernel-trace --memory-copy-trace dir."""
import csv, glob, os, sys, re
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)", n).group(1)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# last tick = events after the last-but-one k_level_hist
starts = [i for i, e in enumerate(ev) if "k_level_hist" in e[2]]
i0 = starts[-2] if len(starts) > 1 else 0
i1 = starts[-1]
t0 = ev[i0][0]
for s, e, n in ev[i0:i1]:
    print(f"{(s - t0) / 1000:9.2f} us  +{(e - s) / 1000:7.2f}  {n}")
