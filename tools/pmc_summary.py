#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "drt::" not in k: continue
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[(short, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:40s} {sum(v)/len(v):18.1f}  (n={len(v)})")
