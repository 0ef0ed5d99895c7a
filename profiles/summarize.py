#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<run>/...) into the small summaries kept under profiles/.

  python profiles/summarize.py stats gpurun_out/r02_stats  > profiles/r02_kernel_stats.md
  python profiles/summarize.py pmc   profiles/r02_hbm_traffic.json gpurun_out/r02_fetch gpurun_out/r02_write > profiles/r02_hbm_traffic.md
  python profiles/summarize.py sq    profiles/r02_valu_pmc.json gpurun_out/r02_sq > profiles/r02_valu_pmc.md
"""
import collections
import csv
import glob
import json
import re
import sys


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    import os
    return max(glob.glob(pattern), key=os.path.getmtime)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def stats(d):
    f = newest(d + "/*/*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | avg us | total ms | % |")
    print("|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                  float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))


def pmc(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = newest(d + "/*/*counter_collection.csv")
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("| kernel | launches | FETCH_SIZE KB/launch (raw) | fetch MB/launch (x2 gfx950 correction) | WRITE_SIZE KB/launch | HBM MB/launch |")
    print("|---|---|---|---|---|---|")
    out = {}
    for k, v in sorted(agg.items()):
        fe = v.get("FETCH_SIZE", [])
        wr = v.get("WRITE_SIZE", [])
        n = max(len(fe), len(wr))
        fkb = sum(fe) / max(len(fe), 1)
        wkb = sum(wr) / max(len(wr), 1)
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 -> double it;
        # WRITE_SIZE is exact.  Both are in KB.
        fetch_mb = 2 * fkb * 1024 / 1e6
        total_mb = fetch_mb + wkb * 1024 / 1e6
        out[k] = total_mb * 1e6
        print("| %s | %d | %.1f | %.3f | %.1f | %.3f |" % (k, n, fkb, fetch_mb, wkb, total_mb))
    return out


def sq(dirs):
    """SQ counters per launch (averaged over the launches of a kernel): instruction counts are per WAVE instruction."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = newest(d + "/*/*counter_collection.csv")
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    print("| kernel | launches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    out = {}
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        row = {c: (sum(v[c]) / len(v[c]) if v.get(c) else None) for c in names}
        out[k] = dict(row, launches=n)
        print("| %s | %d | " % (k, n) + " | ".join("%.4g" % row[c] if row[c] is not None else "-" for c in names) + " |")
    return out


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "sq":
        json.dump(sq(sys.argv[3:]), open(sys.argv[2], "w"), indent=1)
    else:
        json.dump(pmc(sys.argv[3:]), open(sys.argv[2], "w"), indent=1)
