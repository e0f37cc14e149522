#!/bin/bash
# Round-end measurement: PMC traffic of the sampler at the two run shapes (the driver's
# --steps 20 --warmup 3 and the default --steps 5 --warmup 1) into profiles/pmc_traffic.json FIRST, so
# that the bench lines that follow quote the traffic of this very build; then the default bench line
# (with the CPU baseline), the 20-step line, and the rocprofv3 kernel stats of the 20-step command.
# usage: bash scripts/gpu_round3.sh <tag>       (the -m gpu suite: scripts/gpu_full_tests.sh)
tag=${1:-r3f}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/pmc_traffic_$tag.json
bash scripts/gpu_pmc_shape.sh ${tag}a 3 20 || exit 1
cp gpurun_out/pmc_traffic_${tag}a.json gpurun_out/pmc_traffic_$tag.json
cp gpurun_out/pmc_traffic_${tag}a.json gpurun_out/pmc_traffic_${tag}b.json
bash scripts/gpu_pmc_shape.sh ${tag}b 1 5 || exit 1
cp gpurun_out/pmc_traffic_${tag}b.json gpurun_out/pmc_traffic_$tag.json
cp gpurun_out/pmc_traffic_$tag.json profiles/pmc_traffic.json
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
timeout -k 10 500 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_steps20_$tag.json 2> gpurun_out/bench_steps20_$tag.err || { tail -5 gpurun_out/bench_steps20_$tag.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_long_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 --resort-steps 0 > $R/gpurun_out/prof_long_$tag.log 2>&1 || echo "kernel-trace failed"
cd $R
f=$(find gpurun_out/prof_long_$tag -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/kernel_stats_steps20_$tag.csv
find gpurun_out/prof_long_$tag -name '*kernel_trace.csv' -delete
python3 - <<PY
import json
for name in ("bench_$tag.json", "bench_steps20_$tag.json"):
    d = json.load(open("gpurun_out/" + name))
    print(name, "ms/step %.3f" % d["ms_per_step"], "events/s %.4g" % d["value"], "sampler %.3f" % d["roofline"]["avg_launch_ms"],
          "frac %.4f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source"),
          "gather", (d.get("gather_roofline") or {}).get("frac"), "cpu", d["cpu_baseline"]["value"], "match", d["cpu_baseline"].get("gpu_matches_oracle_on_sample"))
PY
grep coag_lane gpurun_out/kernel_stats_steps20_$tag.csv | cut -c1-200
