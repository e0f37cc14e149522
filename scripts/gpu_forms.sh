# The sampler's forms on the shares of the strong-scaling domain (N = 1, 2, 4, 8) and on smaller
# blocks: the library's own choice and each form forced (1: lane per cell with the draw split off,
# 3: lane per cell drawing as it sweeps, 2: its two-wavefront form).  One warm run per size first.
# usage: bash scripts/gpu_forms.sh <tag>
tag=${1:-forms}
mkdir -p gpurun_out
: > gpurun_out/sampler_forms_$tag.txt
for ny in 1 4 13 25 50 100; do
  timeout -k 10 300 python bench.py --ny $ny --no-cpu-baseline --steps 4 --warmup 1 --resort-steps 0 > /dev/null 2>&1
  for lanes in 0 1 3 2; do
    timeout -k 10 300 python bench.py --ny $ny --no-cpu-baseline --steps 8 --warmup 2 --resort-steps 0 --sampler-lanes $lanes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=${lanes} cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'status', d['status_or'])" | tee -a gpurun_out/sampler_forms_$tag.txt
  done
done
