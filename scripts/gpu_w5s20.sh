#!/bin/bash
# the driver's round-end shape (bench.py --gpus 1 --steps 20 --warmup 5): PMC traffic of that shape
# merged into profiles/pmc_traffic.json, then the line itself
tag=${1:-r3h}
mkdir -p gpurun_out
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_$tag.json
bash scripts/gpu_pmc_shape.sh $tag 5 20 || exit 1
cp gpurun_out/pmc_traffic_$tag.json profiles/pmc_traffic.json
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_w5s20_$tag.json 2> gpurun_out/bench_w5s20_$tag.err || { tail -5 gpurun_out/bench_w5s20_$tag.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/bench_w5s20_$tag.json"))
print("ms/step %.3f" % d["ms_per_step"], "events/s %.4g" % d["value"], "sampler %.3f" % d["roofline"]["avg_launch_ms"], "frac %.4f" % d["roofline"]["frac"],
      "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"], "gather", d["gather_roofline"]["frac"], "match", d["cpu_baseline"]["gpu_matches_oracle_on_sample"])
PY
