#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_timestep.py tests/test_gpu_driver.py tests/test_gpu_parity.py -q -k "grow or timestep or driver or scenario or seasalt or emission" > gpurun_out/r4i.log 2>&1; echo "exit $?" >> gpurun_out/r4i.log
tail -6 gpurun_out/r4i.log
timeout -k 10 600 python scripts/fuzz_driver.py 40 7 8 1.5 > gpurun_out/fuzz_driver_h15.txt 2>&1; tail -3 gpurun_out/fuzz_driver_h15.txt
for L in 16 8; do timeout -k 10 300 python bench.py --n-class 10 --steps 2 --warmup 1 --resort-steps 0 --sampler-lanes $L > gpurun_out/bench_ncls10_L$L.json 2> gpurun_out/bench_ncls10_L$L.err; tail -2 gpurun_out/bench_ncls10_L$L.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_ncls10_L$L.json')); print('n_class 10 lanes $L:', 'ms/step %.2f'%d['ms_per_step'], d['kernel_ms_per_step']['coag_sample'], 'status', d['status_or'])"; done
cd /tmp && timeout -k 10 500 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py --steps 150 --trace 30 --given-probs > $GRAFT_REPO_ROOT/gpurun_out/soak_given.json 2> $GRAFT_REPO_ROOT/gpurun_out/soak_given.txt; cd $GRAFT_REPO_ROOT; grep -E "^step" gpurun_out/soak_given.txt | tail -12
python3 -c "
import json; d=json.load(open('gpurun_out/soak_given.json')); w=d['wall_ms_each_step']; q=len(w)//4; print('given probs: wall first quarter %.2f last quarter %.2f'%(sum(w[1:q+1])/q, sum(w[-q:])/q))"
