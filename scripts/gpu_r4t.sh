#!/bin/bash
mkdir -p gpurun_out
for cf in 1.0 1.5 3.0; do
  timeout -k 10 300 python bench.py --nx 50 --ny 50 --n-class 40 --cap-factor $cf --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/bench_cls40_cf$cf.json 2> gpurun_out/bench_cls40_cf$cf.err || { tail -3 gpurun_out/bench_cls40_cf$cf.err; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_cls40_cf$cf.json')); print('n_class 40 cap factor $cf: ms/step %.2f'%d['ms_per_step'], d['kernel_ms_per_step'], 'status', d['status_or'])"
done
