# Cycles per stage of the lane-per-cell sampler's pass (bench.py --sampler-stamps, diagnostic instantiation):
# a rank's share at N = 8 (one wavefront per SIMD: issue + latency of one chain) and the full domain.
for ny in 13 100; do
  timeout -k 10 200 python bench.py --ny $ny --no-cpu-baseline --steps 4 --warmup 1 --resort-steps 0 --sampler-lanes 1 --sampler-stamps 2> gpurun_out/stamps_$ny.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])"
  grep "seq sampler" gpurun_out/stamps_$ny.err
done
