#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_driver.py -q -x > gpurun_out/r4e.log 2>&1; echo "exit $?" >> gpurun_out/r4e.log
tail -40 gpurun_out/r4e.log
bash scripts/gpu_proto.sh r4proto2
