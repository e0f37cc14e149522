# Quick A/B of a sampler build: parity tests of the sampler, then the full-size bench line.
# usage: bash scripts/gpu_try.sh <tag>
tag=${1:-t}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coag or golden or geometry or lane or sampler or config" > gpurun_out/try_$tag.log 2>&1 || { tail -20 gpurun_out/try_$tag.log; exit 1; }
tail -1 gpurun_out/try_$tag.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'])"
