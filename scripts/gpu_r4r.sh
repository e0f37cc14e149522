#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_classes.py tests/test_gpu_driver.py tests/test_gpu_fortran.py tests/test_gpu_timestep.py tests/test_gpu_parity.py tests/test_gpu_grow.py -q -x -k "not lane-per-cell and not lanes-per-cell" > gpurun_out/r4r.log 2>&1; echo "exit $?" >> gpurun_out/r4r.log
tail -6 gpurun_out/r4r.log
grep -q "exit 0" gpurun_out/r4r.log || exit 1
timeout -k 10 600 python scripts/fuzz_transport.py 60 31 > gpurun_out/fuzz_transport_31.log 2>&1; tail -2 gpurun_out/fuzz_transport_31.log
bash scripts/gpu_r4p.sh
