tag=${1:-pre6}
mkdir -p gpurun_out
for ny in 1 2 3 50 100; do
  for lanes in 0 1; do
    timeout -k 10 300 python bench.py --ny $ny --no-cpu-baseline --steps 8 --warmup 2 --resort-steps 0 --sampler-lanes $lanes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=$lanes cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'left', d['sampler_cells_left_to_second_launch'], 'status', d['status_or'])" | tee -a gpurun_out/scale_$tag.txt
  done
done
for nx in 10 30; do
  for lanes in 0 1; do
    timeout -k 10 300 python bench.py --nx $nx --ny 1 --no-cpu-baseline --steps 8 --warmup 2 --resort-steps 0 --sampler-lanes $lanes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nx=$nx ny=1 lanes=$lanes cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])" | tee -a gpurun_out/scale_$tag.txt
  done
done
