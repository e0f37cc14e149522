#!/bin/bash
mkdir -p gpurun_out
bash scripts/gpu_fuzz_all.sh r4n > gpurun_out/fuzz_r4n.log 2>&1; tail -12 gpurun_out/fuzz_r4n.log
for C in 2 10; do
timeout -k 10 400 python bench.py --n-class $C --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/bench_ncls${C}_r4n.json 2> gpurun_out/bench_ncls${C}_r4n.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_ncls${C}_r4n.json')); print('full domain n_class $C:', 'ms/step %.2f'%d['ms_per_step'], 'events/s %.4g'%d['value'], d['kernel_ms_per_step'], 'status', d['status_or'], 'mem', d['device_memory_gb_rank0'])"
done
