# Per-GPU share of the strong-scaling domain on one GPU: the sampler variants at 1/2, 1/4, 1/8
# of the 100x100x30 domain (what each rank holds at N = 2, 4, 8).
for ny in 100 50 25 13; do
  for lanes in 1 8 16; do
    PMCX_SAMPLER_LANES=$lanes timeout -k 10 200 python bench.py --ny $ny --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=$lanes cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])"
  done
done
