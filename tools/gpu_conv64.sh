#!/bin/bash
# correctness of the conv kernels, then the 64-channel layers' microbenchmark
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_realwidth.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 180 --timeout-method thread --tb=short -p no:cacheprovider -k "conv or engine or full or stats" > gpurun_out/conv64_tests.log 2>&1
rc=$?; tail -5 gpurun_out/conv64_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for l in down0.b up3.a down1.a; do
  timeout -k 10 200 python -u tools/conv_microbench.py --iters 10 --only $l 2>&1 | grep "\->" || exit 1
done
