#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms per train step: show_stats.py file.csv [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{r['Name'][:100]:100s} n/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.2f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
print(f"total {tot/1e6/steps:.2f} ms/step")
