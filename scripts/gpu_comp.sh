#!/bin/bash
# components on the device: their tests, the flang tests, and a bench line (does the merge log cost?)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_components.py tests/test_gpu_fortran.py tests/test_gpu_driver.py -q > gpurun_out/comp_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/comp_tests.log
tail -5 gpurun_out/comp_tests.log
