# stage stamps and kernel durations of the sampler with its draw split off
tag=${1:-pre2}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 --sampler-stamps > gpurun_out/stamps_$tag.json 2> gpurun_out/stamps_$tag.err
grep "seq sampler" gpurun_out/stamps_$tag.err
python3 -c "
import json; d=json.load(open('gpurun_out/stamps_$tag.json')); print('stamps build: sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'left', d['sampler_cells_left_to_second_launch'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --resort-steps 0 > $R/gpurun_out/prof_$tag.log 2>&1 || echo "kernel-trace failed"
cd $R
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1); grep -E "coag_lane|Name" "$f" | cut -c1-260
find gpurun_out/prof_$tag -name '*kernel_trace.csv' -delete
