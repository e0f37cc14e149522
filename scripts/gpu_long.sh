# The bench line and the rocprofv3 kernel stats of a longer run (the shape of the driver's
# round-end run: --steps 20 --warmup 3).  usage: bash scripts/gpu_long.sh <tag>
tag=${1:-long}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_long_$tag.json 2> gpurun_out/bench_long_$tag.err || { tail -5 gpurun_out/bench_long_$tag.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_long_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 --resort-steps 0 > $R/gpurun_out/prof_long_$tag.log 2>&1 || echo "kernel-trace failed"
cd $R
f=$(find gpurun_out/prof_long_$tag -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/kernel_stats_long_$tag.csv
find gpurun_out/prof_long_$tag -name '*kernel_trace.csv' -delete
python3 -c "
import json; d=json.load(open('gpurun_out/bench_long_$tag.json')); print('ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'events/s %.4g' % d['value'], 'frac %.4f' % d['roofline']['frac'])"
grep coag_lane gpurun_out/kernel_stats_long_$tag.csv | cut -c1-220
