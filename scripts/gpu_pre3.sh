# the sampler forms on a rank's share of the domain: kernel durations by rocprofv3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ny in 13 25; do
for lanes in 1 2 3; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s${ny}_$lanes -- python3 $R/bench.py --ny $ny --no-cpu-baseline --steps 8 --warmup 2 --resort-steps 0 --sampler-lanes $lanes > $R/gpurun_out/prof_s${ny}_$lanes.log 2>&1 || echo failed
f=$(find $R/gpurun_out/prof_s${ny}_$lanes -name '*kernel_stats.csv' | head -1)
echo "ny=$ny lanes=$lanes"; grep -E "coag_lane" "$f" | cut -d, -f1-4,6,7 | cut -c1-200
find $R/gpurun_out/prof_s${ny}_$lanes -name '*kernel_trace.csv' -delete
done; done
