#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_classes.py tests/test_gpu_parity.py tests/test_gpu_components.py -q -k "grow or classes or weight_preparation or maximum_capacity or overlap or layout or real_case or twenty or column_file" > gpurun_out/r4c.log 2>&1; echo "exit $?" >> gpurun_out/r4c.log
tail -40 gpurun_out/r4c.log
