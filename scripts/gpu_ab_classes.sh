#!/bin/bash
# A/B of library builds (alt_so/*.so) on the sampler with several weight classes, 75 000 cells
mkdir -p gpurun_out
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
: > gpurun_out/ab_classes.txt
for n in base $(ls alt_so/*.so | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $n = base ]; then cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so; else cp alt_so/$n.so wrf-partmc_amd/libpmcx.so; fi
  for C in ${CLASSES:-2 10 40}; do
    timeout -k 10 300 python bench.py --nx 50 --ny 50 --n-class $C --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/abc_${n}_$C.json 2> gpurun_out/abc_${n}_$C.err || { tail -3 gpurun_out/abc_${n}_$C.err; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/abc_${n}_$C.json')); print('$n n_class $C: sampler %.3f'%d['kernel_ms_per_step']['coag_sample'], 'events/step %.5g'%d['events_per_step'], 'status', d['status_or'])" | tee -a gpurun_out/ab_classes.txt
  done
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
