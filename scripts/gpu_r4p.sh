#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_whole_c10 -- python3 $R/scripts/whole_step.py --steps 7 --n-class 10 --trace 3 > $R/gpurun_out/prof_whole_c10.log 2>&1 || { tail -5 $R/gpurun_out/prof_whole_c10.log; exit 1; }
f=$(find $R/gpurun_out/prof_whole_c10 -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print("%-70s calls %5s avg %10.1f us total %10.1f ms" % (r["Name"].replace("pmcx::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
grep "^step" $R/gpurun_out/prof_whole_c10.log | head -12
find $R/gpurun_out/prof_whole_c10 -name '*kernel_trace.csv' -delete
