#!/bin/bash
# Runs the GPU checks in order; a step that ends in a fault / abort / timeout stops the script
# (exit codes other than 0 = pass and 1 = ordinary test failures), so nothing else touches the GPU.
mkdir -p gpurun_out
run() {   # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/ci.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/ci.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name (rc=$rc)"; exit $rc; fi
}
: > gpurun_out/ci.log
for step in "$@"; do
  case $step in
    kernels) run kernels 900 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 120 --timeout-method thread --tb=short -p no:cacheprovider ;;
    optim)   run optim 300 python -u -m pytest tests/test_gpu_optim.py -m gpu -q --timeout 120 --timeout-method thread --tb=short -p no:cacheprovider ;;
    model)   run model 600 python -u -m pytest tests/test_gpu_model.py -m gpu -q --timeout 180 --timeout-method thread --tb=short -p no:cacheprovider ;;
    all)     run all 1100 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread --tb=short -p no:cacheprovider ;;
    smoke)   run smoke 300 python -u -c "import __graft_entry__ as g; g.smoke()" ;;
    micro)   run micro 600 python -u tools/conv_microbench.py --iters 3 ;;
    bench)   run bench 900 python -u bench.py ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
