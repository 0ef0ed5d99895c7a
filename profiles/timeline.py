#!/usr/bin/env python3
"""Print the kernel/copy timeline of the last bench step from a rocprofv3 csv trace directory
(rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py ...)."""
import csv, glob, os, re, sys

d = sys.argv[1]
ev = []
def newest(pattern):      # gpurun merges every call's files into the same directory: take the latest run's
    fs = glob.glob(pattern)
    return [max(fs, key=os.path.getmtime)] if fs else []


for f in newest(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:44]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in newest(d + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]))
ev.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_seed"
idx = [i for i, e in enumerate(ev) if anchor in e[2]]
i0, i1 = idx[-2], idx[-1]
prev = None
busy = 0.0
for e in ev[i0:i1]:
    gap = (e[0] - prev) / 1e3 if prev else 0.0
    print("%8.1f us gap  %8.1f us  %s" % (gap, (e[1] - e[0]) / 1e3, e[2]))
    busy += (e[1] - e[0]) / 1e3
    prev = e[1]
period = (ev[i1][0] - ev[i0][0]) / 1e3
print("step period %.1f us, GPU busy %.1f us, idle %.1f us" % (period, busy, period - busy))
