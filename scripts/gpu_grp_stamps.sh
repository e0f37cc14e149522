#!/bin/bash
mkdir -p gpurun_out
cp wrf-partmc_amd/libpmcx.so /tmp/base.so
for f in alt_so/grp_stamps*.so; do
  cp $f wrf-partmc_amd/libpmcx.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 --sampler-lanes 4 --sampler-stamps > gpurun_out/$(basename $f .so).json 2> gpurun_out/$(basename $f .so).err
  echo "$f"; grep "seq sampler" gpurun_out/$(basename $f .so).err; python3 -c "
import json; d=json.load(open('gpurun_out/$(basename $f .so).json')); print('sampler %.3f'%d['kernel_ms_per_step']['coag_sample'])"
done
cp /tmp/base.so wrf-partmc_amd/libpmcx.so
