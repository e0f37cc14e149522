#!/bin/bash
# transport tests, then the coupled-step profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "transport or trans or lateral or driver or fortran or timestep or slab or config4" > gpurun_out/trans_tests.log 2>&1
rc=$?
echo "tests exit $rc" >> gpurun_out/trans_tests.log
tail -4 gpurun_out/trans_tests.log
[ $rc -eq 0 ] && bash scripts/gpu_whole.sh ${1:-r3c}
