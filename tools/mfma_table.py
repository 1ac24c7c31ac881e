#!/usr/bin/env python3
"""MFMA-pipe utilisation table from the per-layer SQ counter passes of tools/pmc_mfma.sh (gpurun_out/mfma_summary.txt).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel length; kernel length = 4 * SQ_WAVE_CYCLES / SQ_WAVES (every
wave of these kernels lives for the whole launch; SQ_WAVE_CYCLES counts units of 4 shader cycles).
usage: mfma_table.py gpurun_out/mfma_summary.txt"""
import sys
layer, kern, vals, rows = None, None, {}, []


def flush():
    if kern and "SQ_WAVES" in vals:
        length = 4.0 * vals["SQ_WAVE_CYCLES"] / vals["SQ_WAVES"]
        rows.append((layer, kern, length / 1e3, vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / length,
                     vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], vals["SQ_LDS_BANK_CONFLICT"] * 4.0 / 256 / length))


for line in open(sys.argv[1]):
    if line.startswith("== "):
        flush(); kern, vals = None, {}
        layer = line[3:].strip()
    elif not line.startswith(" "):
        flush(); vals = {}
        kern = line.strip()
    else:
        p = line.split()
        vals[p[0]] = float(p[1])
flush()
print("| layer | kernel | kernel length (K cycles) | MFMA busy | wave wait fraction | LDS conflict cycles / CU / cycle |")
print("|---|---|---|---|---|---|")
for l, k, length, busy, wait, conf in rows:
    print(f"| {l} | `{k[:52]}` | {length:.0f} | {busy:.2f} | {wait:.2f} | {conf:.3f} |")
