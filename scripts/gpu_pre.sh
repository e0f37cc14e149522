# The sampler with its draw split off: parity tests of the sampler forms, then the full-size bench
# line in that form and in the one-phase form, and the cells left to the second launch.
# usage: bash scripts/gpu_pre.sh <tag>
tag=${1:-pre}
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -v --timeout 150 -m gpu -k "coag or golden or geometry or split or tombstone or ragged or wide or sort" > gpurun_out/pre_tests_$tag.log 2>&1 || { tail -30 gpurun_out/pre_tests_$tag.log; exit 1; }
tail -2 gpurun_out/pre_tests_$tag.log
for lanes in 0 3 0 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --resort-steps 0 --sampler-lanes $lanes > gpurun_out/bench_${tag}_l$lanes.json 2> gpurun_out/bench_${tag}_l$lanes.err || { tail -5 gpurun_out/bench_${tag}_l$lanes.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_${tag}_l$lanes.json')); print('lanes=$lanes', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'status', d['status_or'])"
done
