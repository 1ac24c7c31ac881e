"""Reads a rocprofv3 kernel-trace CSV of tools/contention_probe.py and prints what ran around the last thief launches
(hardware queue, stream, start relative to the thief, duration)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(collections.Counter((r["Queue_Id"], r["Stream_Id"]) for r in rows))
th = [i for i, r in enumerate(rows) if "thief" in r["Kernel_Name"]]
for i in th[-3:]:
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print(f"--- thief queue {rows[i]['Queue_Id']} stream {rows[i]['Stream_Id']} {(e - s) / 1e3:.0f} us")
    for r in rows:
        rs, re_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if r is not rows[i] and re_ > s - 100000 and rs < e + 100000:
            print(f"  q{r['Queue_Id']} s{r['Stream_Id']} {(rs - s) / 1e3:9.1f} {(re_ - rs) / 1e3:9.1f}  {re.sub(r'[(].*', '', r['Kernel_Name'])[:64]}")
