#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of the coupled step (each counter its own pass)
tag=${1:-wp}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcw_${tag}_$pass -- python3 $R/scripts/whole_step.py --steps 4 > $R/gpurun_out/pmcw_${tag}_$pass.log 2>&1 || { echo "pass $pass failed"; tail -5 $R/gpurun_out/pmcw_${tag}_$pass.log; exit 1; }
done
cd $R
python3 - <<PY
import csv, glob, collections
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmcw_${tag}_%s/**/*counter_collection.csv" % counter, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                d[row["Kernel_Name"].split("(")[0].replace("pmcx::", "").replace("void ", "")].append(float(row["Counter_Value"]))
    for k in ("k_trans_decide", "k_trans_gather", "k_trans_remove", "k_dry_deposition", "k_scenario_update", "k_bin_sort"):
        if k in d:
            print(counter, k, " ".join("%.0f" % (v / 1024.0) for v in d[k]), "MiB per call")
PY
find gpurun_out -name '*counter_collection.csv' -size +5M -delete
