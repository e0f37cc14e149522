#!/bin/bash
# the whole -m gpu suite, output under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/full_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/full_tests.log
tail -6 gpurun_out/full_tests.log
