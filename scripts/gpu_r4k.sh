#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_classes.py tests/test_gpu_parity.py tests/test_gpu_components.py tests/test_gpu_species20.py -q -x > gpurun_out/r4k.log 2>&1; echo "exit $?" >> gpurun_out/r4k.log
tail -8 gpurun_out/r4k.log
grep -q "exit 0" gpurun_out/r4k.log || exit 1
bash scripts/gpu_classes_sweep.sh
