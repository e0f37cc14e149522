#!/bin/bash
# A/B of library builds on the coupled step (scripts/whole_step.py): the in-tree library and each alternative
# under alt_so/, interleaved REPS times; device ms per launch of the transport (decide + room + gather +
# remove + compact) and wall ms of the steady non-chemistry steps.  usage: bash scripts/gpu_ab_whole.sh <tag>
tag=${1:-abw}; reps=${REPS:-3}
mkdir -p gpurun_out
: > gpurun_out/abw_$tag.txt
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
run() {
  (cd /tmp && timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py --steps 7 > $GRAFT_REPO_ROOT/gpurun_out/abw_${tag}_$name.json 2> $GRAFT_REPO_ROOT/gpurun_out/abw_${tag}_$name.err) || { tail -5 gpurun_out/abw_${tag}_$name.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/abw_${tag}_$name.json')); w=d['wall_ms_each_step']; print('$name', '%.3f'%d['device_ms']['transport']['per_launch'], '%.2f'%min(w[1], w[2], w[4], w[5]))" >> gpurun_out/abw_$tag.txt
}
name=warm; run
for r in $(seq $reps); do
  cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so; name=base; run
  for f in $(ls alt_so/*.so 2>/dev/null); do name=$(basename $f .so); cp $f wrf-partmc_amd/libpmcx.so; run; done
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
cat gpurun_out/abw_$tag.txt
