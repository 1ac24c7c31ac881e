#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 counter_collection CSVs: pmc_summary.py <dir> [name-filter]"""
import csv, glob, collections, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if flt and flt not in k:
        continue
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {sum(x)/len(x):16.1f}  (n={len(x)})")
