# Per-GPU share of the strong-scaling domain on one GPU: what each rank holds at N = 1, 2, 4, 8
# (100x100x30 cells split by columns), with the library's own choice of sampler and, for
# comparison, each variant forced (1: lane per cell, 2: two-wavefront form, 8: 8 lanes per cell).
for ny in 100 50 25 13; do
  for lanes in "" 1 2 8; do
    timeout -k 10 200 python bench.py --ny $ny --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 --sampler-lanes ${lanes:-0} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=${lanes:-auto} cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])"
  done
done
