#!/bin/bash
# kernel trace of the conv microbenchmark for one layer: resource columns + durations
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_micro
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_micro -- python3 $R/tools/conv_microbench.py --iters 3 --only ${1:-down0.b} > $R/gpurun_out/prof_micro.log 2>&1
find $R/gpurun_out/prof_micro -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/prof_micro_trace.csv \;
rm -rf $R/gpurun_out/prof_micro
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_micro_trace.csv")))
for r in rows:
    if "conv3x3" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:50], "lds", r["LDS_Block_Size"], "scratch", r["Scratch_Size"], "vgpr", r["VGPR_Count"], "agpr", r["Accum_VGPR_Count"], "sgpr", r["SGPR_Count"], "wg", r["Workgroup_Size_X"], "grid", r["Grid_Size_X"], "us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
PY
