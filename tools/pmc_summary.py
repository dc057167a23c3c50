#!/usr/bin/env python3
"""Counter CSVs of several `rocprofv3 --pmc` passes -> per counter: the median over the launches of the early cull kernel
(launches after the first; the first sees a cleared HZB).   python tools/pmc_summary.py <outdir> <dir prefix> [label]"""
import csv, glob, sys, collections, statistics
out, prefix = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{prefix}*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in res:
    if "meshletCullKernel" in k and ("true, true, true, true" in k):
        print(label, k.replace("(anonymous namespace)::", "")[:70])
        for c in sorted(res[k]):
            v = res[k][c]
            print(f"   {c:40s} first {v[0]:14.0f}  median of the rest {statistics.median(v[1:] or v):14.0f}  ({len(v)} launches)")
