# FETCH_SIZE and WRITE_SIZE of the pair sampler at ONE run shape of bench.py (each counter its own
# rocprofv3 --pmc pass, nothing else traced), summarised per launch of the TIMED steps by
# scripts/pmc_shape.py.  usage: bash scripts/gpu_pmc_shape.sh <tag> <warmup> <steps> [bench args...]
tag=$1; W=$2; K=$3; shift 3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcs_${tag}_$pass -- python3 $R/bench.py --no-cpu-baseline --resort-steps 0 --warmup $W --steps $K "$@" > $R/gpurun_out/pmcs_${tag}_$pass.log 2>&1 || { echo "pass $pass failed"; tail -5 $R/gpurun_out/pmcs_${tag}_$pass.log; exit 1; }
done
cd $R
python3 scripts/pmc_shape.py gpurun_out $tag $W $K gpurun_out/pmc_traffic_$tag.json
