#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py -q -k "grouped" -x > gpurun_out/grp_tests.log 2>&1; echo "exit $?" >> gpurun_out/grp_tests.log
tail -25 gpurun_out/grp_tests.log
grep -q "exit 0" gpurun_out/grp_tests.log || exit 1
for L in 0 4 0 4; do timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --resort-steps 0 --sampler-lanes $L > gpurun_out/grp_bench_L$L.json 2> gpurun_out/grp_bench_L$L.err || { tail -5 gpurun_out/grp_bench_L$L.err; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/grp_bench_L$L.json')); print('lanes $L:', 'sampler %.3f'%d['kernel_ms_per_step']['coag_sample'], 'ms/step %.3f'%d['ms_per_step'], 'events/step %.5g'%d['events_per_step'], 'status', d['status_or'], 'left', d['sampler_cells_left_to_second_launch'])"; done
