mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1 || true
tail -3 gpurun_out/t5.log
PMCX_SAMPLER_LANES=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mc_coag_parity or ragged or dense" 2>&1 | tail -3
for cfg in "8 2 3 128" "1 1 3 32" "1 1 2 32" "1 1 3 16" "1 2 3 16" "1 1 4 16"; do
  set -- $cfg
  PMCX_SAMPLER_LANES=$1 PMCX_SAMPLER_WPB=$2 PMCX_SAMPLER_OCC=$3 timeout -k 10 200 python bench.py --nx 50 --ny 50 --steps 3 --warmup 1 --max-appends $4 --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('L=$1 wpb=$2 occ=$3 app=$4', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['status_or'])"
done
