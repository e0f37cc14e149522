#!/bin/bash
# the pair sampler with the reference's 20 species (em_scm_xy/aero_data.dat) at the shares of the domain that fit
# one GPU's memory: 50 x 50 x 30 (the N = 4 share) and 50 x 25 x 30 (N = 8)
mkdir -p gpurun_out
for shape in "50 50" "50 25"; do
  set -- $shape
  timeout -k 10 500 python bench.py --nx $1 --ny $2 --n-spec 20 --no-cpu-baseline --steps 10 --warmup 3 --resort-steps 0 > gpurun_out/bench_spec20_$1x$2.json 2> gpurun_out/bench_spec20_$1x$2.err || { tail -5 gpurun_out/bench_spec20_$1x$2.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_spec20_$1x$2.json')); print('$1 x $2 x 30, 20 species: ms/step %.3f'%d['ms_per_step'], 'events/s %.4g'%d['value'], 'frac %.4f'%d['roofline']['frac'], d['kernel_ms_per_step'], 'status', d['status_or'])"
done
