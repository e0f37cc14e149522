#!/bin/bash
# PMC counters of k_trans_decide in the coupled step, one class against ten (each counter its own pass)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in 1 10; do
for c in VALUBusy SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS MeanOccupancyPerCU; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcd_${C}_$c -- python3 $R/scripts/whole_step.py --steps 4 --n-class $C > $R/gpurun_out/pmcd_${C}_$c.log 2>&1 || { echo "pass $C $c failed"; tail -2 $R/gpurun_out/pmcd_${C}_$c.log; }
done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcd_*")):
    if d.endswith(".log"): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pmcx::", "")
            if k.startswith(("k_trans_decide", "k_trans_gather")):
                acc[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(d.rsplit("/", 1)[1], k, c, ["%.4g" % x for x in v])
PY
find gpurun_out -name '*counter_collection.csv' -size +5M -delete
