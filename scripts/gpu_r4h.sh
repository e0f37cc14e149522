#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_gpu_fortran.py -q -k "grow or transport or driver or fortran" > gpurun_out/r4h.log 2>&1; echo "exit $?" >> gpurun_out/r4h.log
tail -6 gpurun_out/r4h.log
timeout -k 10 600 python scripts/fuzz_driver.py 40 7 8 1.5 > gpurun_out/fuzz_driver_h15.txt 2>&1; tail -3 gpurun_out/fuzz_driver_h15.txt
cd /tmp && timeout -k 10 500 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py --steps 150 --trace 15 > $GRAFT_REPO_ROOT/gpurun_out/soak_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/soak_trace.txt; cd $GRAFT_REPO_ROOT; grep -E "^step" gpurun_out/soak_trace.txt | tail -30
python3 -c "
import json; d=json.load(open('gpurun_out/soak_trace.json')); w=d['wall_ms_each_step']; q=len(w)//4; print('wall first quarter %.2f last quarter %.2f'%(sum(w[1:q+1])/q, sum(w[-q:])/q))"
