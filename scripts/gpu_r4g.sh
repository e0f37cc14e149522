#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/full_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/full_tests.log
tail -6 gpurun_out/full_tests.log
bash scripts/gpu_round4.sh r4a
timeout -k 10 600 python scripts/fuzz_driver.py 24 7 8 1.5 > gpurun_out/fuzz_driver_h15.txt 2>&1; tail -4 gpurun_out/fuzz_driver_h15.txt
