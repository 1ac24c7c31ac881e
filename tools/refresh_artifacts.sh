#!/bin/bash
# One GPU-box pass that regenerates every measurement artefact of the current build (run via gpurun from the repo root):
#   bash tools/refresh_artifacts.sh <tag>
# kernel-trace stats of the bench step, HBM traffic (two PMC passes), MFMA-pipe utilisation per layer (SQ pass per
# layer over the microbenchmark) and the bench line itself.  Every step runs under its own time limit; the first
# failure stops the script.  Results: gpurun_out/<...>; copy what is to be kept into profiles/ afterwards.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_bench.sh $TAG
bash $R/tools/pmc_bench.sh
rm -f $R/gpurun_out/mfma_summary.txt
bash $R/tools/pmc_mfma.sh down0.b down1.b down2.b down3.b bott.b up0.a up3.a > /dev/null
cd $R
timeout -k 10 600 python3 -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json
