#!/bin/bash
# per-kernel times of the sampler with several weight classes (75 000 cells)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in ${CLASSES:-2 10 40}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cls_$C -- python3 $R/bench.py --nx 50 --ny 50 --n-class $C --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 > $R/gpurun_out/prof_cls_$C.log 2>&1 || { tail -5 $R/gpurun_out/prof_cls_$C.log; exit 1; }
  f=$(find $R/gpurun_out/prof_cls_$C -name '*kernel_stats.csv' | head -1)
  echo "== n_class $C"
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:8]:
    print("%-60s calls %5s avg %10.1f us total %10.1f ms" % (r["Name"].replace("pmcx::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
  find $R/gpurun_out/prof_cls_$C -name '*kernel_trace.csv' -delete
done
