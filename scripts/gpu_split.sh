# Where the two-wavefront form of the lane-per-cell sampler pays: small blocks of cells.
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split" > gpurun_out/t.log 2>&1 || { tail -15 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for ny in 1 4 8 13 15 18 21; do
  for lanes in "" 1 2 8; do
    timeout -k 10 120 python bench.py --ny $ny --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 --sampler-lanes ${lanes:-0} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=${lanes:-auto} cells', d['config']['cells'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])" || exit 1
  done
done
