#!/bin/bash
# per-kernel statistics of a longer coupled run (populations and weights have drifted apart)
tag=${1:-soak}; steps=${2:-90}; shift 2 2>/dev/null
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_soak_$tag -- python3 $R/scripts/whole_step.py --steps $steps "$@" > $R/gpurun_out/soak_prof_$tag.log 2>&1 || { tail -5 $R/gpurun_out/soak_prof_$tag.log; exit 1; }
f=$(find $R/gpurun_out/prof_soak_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/soak_kernel_stats_$tag.csv
t=$(find $R/gpurun_out/prof_soak_$tag -name '*kernel_trace.csv' | head -1)
python3 - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$t")):
    n = r["Kernel_Name"].split("(")[0].replace("pmcx::", "").replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        q = len(v) // 4
        print("%-28s calls %4d  first quarter avg %8.0f us   last quarter avg %8.0f us" % (n[:28], len(v), sum(v[1:q + 1]) / q, sum(v[-q:]) / q))
PY
find $R/gpurun_out/prof_soak_$tag -name '*kernel_trace.csv' -delete
