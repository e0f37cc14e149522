#!/bin/bash
# PMC evidence for round 4's new kernels (each counter its own rocprofv3 --pmc pass, nothing else traced):
#   k_vol_merge, 20 species, 75 000 cells: FETCH_SIZE, WRITE_SIZE per launch against 480 B per event
#   k_coag_screen, 40 classes, 75 000 cells: VALU instructions and busy cycles
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, counter, bench args
  local n=$1 c=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcn_${n}_$c -- python3 $R/bench.py --no-cpu-baseline --resort-steps 0 --warmup 1 --steps 3 "$@" > $R/gpurun_out/pmcn_${n}_$c.log 2>&1 || { echo "pass $n $c failed"; tail -3 $R/gpurun_out/pmcn_${n}_$c.log; }
}
for c in FETCH_SIZE WRITE_SIZE; do run spec20 $c --nx 50 --ny 50 --n-spec 20; done
for c in SQ_INSTS_VALU SQ_BUSY_CYCLES VALUBusy GRBM_GUI_ACTIVE; do run cls40 $c --nx 50 --ny 50 --n-class 40; done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("gpurun_out/pmcn_*")):
    if not d.rsplit("/", 1)[1].count("_") >= 2 or d.endswith(".log"): continue
    name = d.rsplit("/", 1)[1][5:]
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pmcx::", "")
            if k.startswith(("k_vol_merge", "k_coag_screen", "k_coag_lane", "k_bin_sort", "k_rebalance")):
                acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        out[f"{name} | {k} | {c}"] = {"launches": len(v), "mean": sum(v) / len(v), "last": v[-1]}
for k, v in out.items(): print(k, v)
json.dump(out, open("gpurun_out/pmc_new_kernels.json", "w"), indent=1)
PY
for f in gpurun_out/pmcn_spec20_FETCH_SIZE.log gpurun_out/pmcn_cls40_SQ_INSTS_VALU.log; do grep '"metric"' $f | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['config']['workload'], 'events/step', d['events_per_step'], 'sampler ms', d['kernel_ms_per_step']['coag_sample'])"; done
find gpurun_out -name '*counter_collection.csv' -size +5M -delete
