#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_driver.py -q > gpurun_out/r4f.log 2>&1; echo "exit $?" >> gpurun_out/r4f.log
tail -15 gpurun_out/r4f.log
bash scripts/gpu_proto.sh r4proto3
bash scripts/gpu_soak_stats.sh r4soak_a 150 > gpurun_out/r4soak_a.txt 2>&1; tail -22 gpurun_out/r4soak_a.txt
bash scripts/gpu_soak_stats.sh r4soak_b 150 --cap-factor 3.0 > gpurun_out/r4soak_b.txt 2>&1; tail -22 gpurun_out/r4soak_b.txt
timeout -k 10 400 python bench.py --n-class 10 --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/bench_ncls10.json 2> gpurun_out/bench_ncls10.err; tail -3 gpurun_out/bench_ncls10.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_ncls10.json')); print('n_class 10:', 'ms/step %.2f'%d['ms_per_step'], 'events/s %.4g'%d['value'], d['kernel_ms_per_step'], 'status', d['status_or'], 'mem GB %.1f'%d['device_memory_gb_rank0'])"
