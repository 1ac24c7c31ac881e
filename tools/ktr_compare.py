"""Per-launch comparison of two rocprofv3 kernel traces of tools/contention_probe.py (same build, same steps): base run
against a run with the thief.  Launches of the compute streams are matched by position; prints, per kernel name, the summed
extra time of the launches of the LAST step, and the launches that grew by more than 20 us."""
import collections
import csv
import re
import sys


def load(path):
    rows = [r for r in csv.DictReader(open(path)) if "thief" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def name(r):
    return re.sub(r"[(].*", "", r["Kernel_Name"]).replace("void ", "")[:72]


a, b = load(sys.argv[1]), load(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
assert len(a) == len(b), (len(a), len(b))
per = len(a) // steps
extra = collections.defaultdict(float)
base = collections.defaultdict(float)
big = []
for ra, rb in zip(a[-per:], b[-per:]):
    assert name(ra) == name(rb), (name(ra), name(rb))
    da = (int(ra["End_Timestamp"]) - int(ra["Start_Timestamp"])) / 1e3
    db = (int(rb["End_Timestamp"]) - int(rb["Start_Timestamp"])) / 1e3
    extra[name(ra)] += db - da
    base[name(ra)] += da
    if db - da > 20:
        big.append((name(ra), da, db))
print(f"launches per step ~{per}; sum of kernel time base {sum(base.values()) / 1e3:.2f} ms, extra {sum(extra.values()) / 1e3:+.2f} ms")
for k, v in sorted(extra.items(), key=lambda kv: -abs(kv[1]))[:12]:
    print(f"  {v:+9.1f} us of {base[k]:9.1f}  {k}")
print("launches that grew by > 20 us:")
for n, da, db in big:
    print(f"  {da:8.1f} -> {db:8.1f}  {n}")
