#!/bin/bash
# MFMA-pipe utilisation of the conv / wgrad kernels per layer: one SQ counter pass per layer over the microbenchmark
# (counters only, no trace domains).  tools/pmc_mfma.sh <layer> [<layer> ...]; summary with tools/pmc_summary.py.
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for LAYER in "$@"; do
  O=$R/gpurun_out/mfma_$LAYER
  rm -rf $O; mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O -- python3 $R/tools/conv_microbench.py --only $LAYER --iters 2 > $O.log 2>&1
  echo "== $LAYER" >> $R/gpurun_out/mfma_summary.txt
  python3 $R/tools/pmc_summary.py $O conv >> $R/gpurun_out/mfma_summary.txt
done
tail -60 $R/gpurun_out/mfma_summary.txt
