#!/bin/bash
# PMC passes over the conv microbenchmark (one layer), each pass in its own run (no trace domains).
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
LAYER=${1:-down0.b}
O=$R/gpurun_out/pmc_$LAYER
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/conv_microbench.py --only $LAYER --iters 2 > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/fetch -- python3 $R/tools/conv_microbench.py --only $LAYER --iters 2 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/write -- python3 $R/tools/conv_microbench.py --only $LAYER --iters 2 > $O/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -- python3 $R/tools/conv_microbench.py --only $LAYER --iters 2 > $O/sq2.log 2>&1
find $O -name "*counter_collection.csv" | head
