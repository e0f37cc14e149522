# Kernel-level profile of the coupled step (scripts/whole_step.py) at its default size.
# usage: bash scripts/gpu_whole.sh <tag>
tag=${1:-w}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python3 $R/scripts/whole_step.py --steps 6 > $R/gpurun_out/whole_$tag.json 2> $R/gpurun_out/whole_$tag.err || { tail -5 $R/gpurun_out/whole_$tag.err; exit 1; }
cat $R/gpurun_out/whole_$tag.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_whole_$tag -- python3 $R/scripts/whole_step.py --steps 6 > $R/gpurun_out/whole_prof_$tag.log 2>&1 || { tail -5 $R/gpurun_out/whole_prof_$tag.log; exit 1; }
f=$(find $R/gpurun_out/prof_whole_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/whole_kernel_stats_$tag.csv
t=$(find $R/gpurun_out/prof_whole_$tag -name '*kernel_trace.csv' | head -1)
python3 - <<PY > $R/gpurun_out/whole_per_call_$tag.txt
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$t")):
    n = r["Kernel_Name"].split("(")[0].replace("pmcx::", "").replace("void ", "").split("<")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n in ("k_trans_decide", "k_trans_gather", "k_trans_remove", "k_dry_deposition", "k_scenario_update", "k_bin_sort", "k_kernel_bounds"):
    if n in d:
        print("%-20s us per call: %s" % (n, " ".join("%.0f" % x for x in d[n])))
PY
cat $R/gpurun_out/whole_per_call_$tag.txt
find $R/gpurun_out/prof_whole_$tag -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/whole_kernel_stats_$tag.csv")))
for r in rows[:16]:
    print("%-60s calls %5s avg %10.1f us total %9.1f ms %5s%%"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
