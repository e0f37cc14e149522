# Instruction-cache counters of the sampler (separate --pmc pass, as the pool requires).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_" | head -40 > $R/gpurun_out/icache_counters.txt
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_ic_$name -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --resort-steps 0 > $R/gpurun_out/pmc_ic_$name.log 2>&1 || echo "pass $pass failed"
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out ic | grep -i "coag_lane"
