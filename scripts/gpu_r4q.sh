#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_classes.py tests/test_gpu_driver.py tests/test_gpu_fortran.py tests/test_gpu_timestep.py tests/test_gpu_parity.py -q -x -k "not lane-per-cell and not lanes-per-cell" > gpurun_out/r4q.log 2>&1; echo "exit $?" >> gpurun_out/r4q.log
tail -6 gpurun_out/r4q.log
grep -q "exit 0" gpurun_out/r4q.log || exit 1
bash scripts/gpu_r4p.sh
