#!/bin/bash
# the pair sampler against the number of weight classes (75 000 cells x 10^4, one species)
mkdir -p gpurun_out
for C in 1 2 3 5 8 10 20 40; do
  timeout -k 10 300 python bench.py --nx 50 --ny 50 --n-class $C --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/bench_cls_$C.json 2> gpurun_out/bench_cls_$C.err || { tail -5 gpurun_out/bench_cls_$C.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_cls_$C.json')); print('n_class $C: ms/step %.3f'%d['ms_per_step'], 'sampler %.3f'%d['kernel_ms_per_step']['coag_sample'], 'rebalance %.3f'%d['kernel_ms_per_step']['rebalance'], 'events/step %.4g'%d['events_per_step'], 'cand/s %.4g'%d['candidate_pairs_per_sec'], 'status', d['status_or'], 'left', d['sampler_cells_left_to_second_launch'])"
done
