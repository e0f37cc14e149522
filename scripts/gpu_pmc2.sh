# FETCH/WRITE + SQ PMC passes of the sampler in one call (each its own rocprofv3 run).
# usage: bash scripts/gpu_pmc2.sh <tag> [bench args...]
mkdir -p gpurun_out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_${tag}_$name.log 2>&1 || echo "pass $pass failed"
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out $tag | grep -E "coag|bin_sort"
