#!/bin/bash
# the pair sampler's duration launch by launch at the driver's shape (rocprofv3 --kernel-trace), beside the bench line's own
# HIP-event average over the timed launches: the --stats summary averages the warm-up launches in
tag=${1:-r4}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sl_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --resort-steps 0 > $R/gpurun_out/prof_sl_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_sl_$tag.log; exit 1; }
cd $R
t=$(find gpurun_out/prof_sl_$tag -name '*kernel_trace.csv' | head -1)
python3 - <<PY
import csv, json
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in csv.DictReader(open("$t")) if "k_coag_lane" in r["Kernel_Name"]]
d.sort(); ms = [x for _, x in d]
line = None
for ln in open("gpurun_out/prof_sl_$tag.log"):
    if ln.startswith("{") and '"metric"' in ln: line = json.loads(ln)
out = {"kernel": "k_coag_lane<false, true, 4, false, false, 24, false>", "launches_ms": [round(x, 4) for x in ms],
       "mean_all": sum(ms) / len(ms), "mean_timed_last_20": sum(ms[-20:]) / 20,
       "bench_line_avg_launch_ms": line["roofline"]["avg_launch_ms"], "bench_line_frac": line["roofline"]["frac"],
       "bench_line_ms_per_step": line["ms_per_step"], "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"]}
out["frac_from_trace_timed"] = out["algorithmic_bytes_per_launch"] / (out["mean_timed_last_20"] * 1e-3) / 8e12
json.dump(out, open("gpurun_out/sampler_launches_$tag.json", "w"), indent=1)
print(json.dumps(out))
PY
find gpurun_out/prof_sl_$tag -name '*kernel_trace.csv' -delete
