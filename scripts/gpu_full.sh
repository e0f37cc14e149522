# Round-end measurement: tests, full-size bench, rocprofv3 kernel stats, PMC traffic.
# usage: bash scripts/gpu_full.sh <tag>
tag=${1:-rX}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/tests_$tag.log 2>&1; tail -3 gpurun_out/tests_$tag.log
timeout -k 10 500 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $R/gpurun_out/prof_$tag.log 2>&1 || echo "kernel-trace failed"
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_${tag}_$pass -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_${tag}_$pass.log 2>&1 || echo "pass $pass failed"
done
cd $R
python3 scripts/pmc_summary.py gpurun_out $tag > gpurun_out/pmc_${tag}_summary.txt
python3 - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print(json.dumps({k:d[k] for k in ["value","ms_per_step","kernel_ms_per_step","roofline","particle_steps_per_sec","gpu_vs_cpu_events_per_sec","status_or"]},indent=1))
print(d["cpu_baseline"])
PY
cat gpurun_out/pmc_${tag}_summary.txt | cut -c1-400
cat $(find gpurun_out/prof_$tag -name "*kernel_stats.csv") | cut -c1-160
