#!/bin/bash
# Round-4 measurement set on one box: PMC traffic of the sampler at the driver's shape (--steps 20 --warmup 5)
# into profiles/pmc_traffic.json FIRST, the bench line of that shape (quoting that traffic), the default line
# (with the CPU leg), the rocprofv3 kernel statistics of the driver's shape, the coupled step, and the
# side measurement with ten weight classes.  usage: bash scripts/gpu_round4.sh <tag>
tag=${1:-r4a}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash scripts/gpu_w5s20.sh $tag || exit 1
timeout -k 10 500 python bench.py > gpurun_out/bench_default_$tag.json 2> gpurun_out/bench_default_$tag.err || { tail -5 gpurun_out/bench_default_$tag.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w5s20_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --resort-steps 0 > $R/gpurun_out/prof_w5s20_$tag.log 2>&1 || echo "kernel-trace failed"
cd $R
f=$(find gpurun_out/prof_w5s20_$tag -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/kernel_stats_w5s20_$tag.csv
find gpurun_out/prof_w5s20_$tag -name '*kernel_trace.csv' -delete
grep coag_lane gpurun_out/kernel_stats_w5s20_$tag.csv | cut -c1-220
python3 -c "
import json; d=json.load(open('gpurun_out/bench_default_$tag.json')); print('default: ms/step %.3f'%d['ms_per_step'], 'events/s %.4g'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'cpu %.4g'%d['cpu_baseline']['value'], 'x%.0f'%d['gpu_vs_cpu_events_per_sec'], 'match', d['cpu_baseline']['gpu_matches_oracle_on_sample'])"
timeout -k 10 400 python bench.py --n-class 10 --steps 3 --warmup 1 --resort-steps 0 > gpurun_out/bench_ncls10_$tag.json 2> gpurun_out/bench_ncls10_$tag.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_ncls10_$tag.json')); print('n_class 10:', 'ms/step %.2f'%d['ms_per_step'], 'events/s %.4g'%d['value'], d['kernel_ms_per_step'], 'status', d['status_or'])"
bash scripts/gpu_whole.sh $tag > gpurun_out/whole_$tag.txt 2>&1; tail -30 gpurun_out/whole_$tag.txt
