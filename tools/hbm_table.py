#!/usr/bin/env python3
"""Achieved HBM-side GB/s per kernel: PMC bytes per launch (profiles/r01_pmc_traffic.json, tools/pmc_bench.sh)
divided by the average duration of the same kernel in a rocprofv3 --kernel-trace --stats summary of the same
bench command.  usage: hbm_table.py <traffic.json> <kernel_stats.csv> [min_ms_total]"""
import csv, json, sys
traffic = json.load(open(sys.argv[1]))["per_kernel"]
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
rows = []
for name, t in traffic.items():
    s = stats.get(name)
    if s is None:
        continue
    avg_ns = float(s["AverageNs"]); calls = int(s["Calls"])
    gb = t["hbm_bytes_per_launch"] / 1e9
    rows.append((avg_ns * calls, name, calls, avg_ns / 1e3, gb, gb / (avg_ns * 1e-9) if avg_ns else 0.0))
rows.sort(reverse=True)
print("| kernel | launches | avg µs | GB per launch (PMC) | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for tot, name, calls, us, gb, gbps in rows:
    if tot < float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else False:
        continue
    short = name.split("(")[0].replace("void ", "")[:48]
    print(f"| `{short}` | {calls} | {us:.0f} | {gb:.3f} | {gbps:.0f} | {gbps / 8000:.2f} |")
