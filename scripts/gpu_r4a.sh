#!/bin/bash
# round 4, first call: the new tests (law 1 overlap, many classes), then the whole -m gpu suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_classes.py -q -x > gpurun_out/r4a_classes.log 2>&1; echo "classes exit $?" >> gpurun_out/r4a_classes.log
tail -15 gpurun_out/r4a_classes.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "transport or weight_preparation" > gpurun_out/r4a_trans.log 2>&1; echo "transport exit $?" >> gpurun_out/r4a_trans.log
tail -15 gpurun_out/r4a_trans.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/full_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/full_tests.log
tail -30 gpurun_out/full_tests.log
