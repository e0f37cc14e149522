# parity tests of the sampler forms, then the forms on the shares of N = 1, 2, 4, 8
tag=${1:-pre4}
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout 150 -m gpu -k "coag or golden or geometry or split or tombstone or ragged or wide or sort" > gpurun_out/pre_tests_$tag.log 2>&1 || { tail -30 gpurun_out/pre_tests_$tag.log; exit 1; }
tail -2 gpurun_out/pre_tests_$tag.log
for ny in 13 25 50 100; do
  for lanes in 1 3 2; do
    timeout -k 10 300 python bench.py --ny $ny --no-cpu-baseline --steps 8 --warmup 2 --resort-steps 0 --sampler-lanes $lanes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ny=$ny lanes=$lanes cells', d['config']['cells'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'left', d['sampler_cells_left_to_second_launch'], 'status', d['status_or'])" | tee -a gpurun_out/scale_$tag.txt
  done
done
