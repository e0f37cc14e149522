mkdir -p gpurun_out
for W in 8 16 32; do
  cp wrf-partmc_amd/libpmcx_w$W.so wrf-partmc_amd/libpmcx.so
  for L in 8 16; do
  PMCX_SAMPLER_LANES=$L timeout -k 10 200 python bench.py --nx 50 --ny 50 --steps 3 --warmup 1 --max-appends 128 --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('WIN=$W L=$L', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['status_or'])"
  done
done
