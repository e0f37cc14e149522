#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_classes.py -q > gpurun_out/r4d.log 2>&1; echo "exit $?" >> gpurun_out/r4d.log
tail -25 gpurun_out/r4d.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/full_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/full_tests.log
tail -12 gpurun_out/full_tests.log
bash scripts/gpu_proto.sh r4proto
