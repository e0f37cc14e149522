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
find $R/gpurun_out/prof_whole_$tag -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/whole_kernel_stats_$tag.csv")))
for r in rows[:16]:
    print("%-60s calls %5s avg %10.1f us total %9.1f ms %5s%%"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
