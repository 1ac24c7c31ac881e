#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two PMC passes of tools/pmc_bench.sh.
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced read, so it is doubled; WRITE_SIZE is exact."""
import csv, glob, hashlib, json, os, sys, collections
d = sys.argv[1]
FILTER = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_v2_kernel"       # substring(s, ";"-separated: ALL must occur) of the kernels averaged into traffic.json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.environ.get("UNET_HIP_LIB") or os.path.join(ROOT, "unet_amd", "libunet_hip.so"), "rb") as f:
    BUILD = hashlib.sha256(f.read()).hexdigest()[:12]      # the same id bench.py prints as "build"
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"]) - sum(kv[1]["WRITE_SIZE"])):
    nf, nw = len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"])
    if not nf or not nw:
        continue
    rd = 2.0 * 1024 * sum(v["FETCH_SIZE"]) / nf
    wr = 1024 * sum(v["WRITE_SIZE"]) / nw
    out[k] = {"launches": nf, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    print(f"{k[:70]:70s} n={nf:4d} read {rd/1e6:9.1f} MB  write {wr/1e6:9.1f} MB per launch")
conv = [v for k, v in out.items() if all(f in k for f in FILTER.split(";"))]
if conv:
    n = sum(c["launches"] for c in conv)
    tot = sum(c["launches"] * c["hbm_bytes_per_launch"] for c in conv)
    json.dump({"kernel": f"kernels matching {FILTER!r}" if FILTER != "conv3x3_v2_kernel" else "conv3x3_v2_kernel (all instantiations)", "launches": n, "hbm_bytes_per_launch": tot / n, "build": BUILD,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; 2*FETCH_SIZE + WRITE_SIZE, KiB units",
               "per_kernel": out}, open(f"{d}/traffic.json", "w"), indent=1)
    print(f"{FILTER} average HBM bytes per launch: {tot/n/1e6:.1f} MB over {n} launches")
