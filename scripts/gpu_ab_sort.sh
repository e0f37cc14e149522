#!/bin/bash
# A/B of library builds (alt_so/*.so): k_bin_sort in the coupled step (one class)
mkdir -p gpurun_out
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
cd /tmp
for r in 1 2; do
for n in base $(ls $GRAFT_REPO_ROOT/alt_so/*.so | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $n = base ]; then cp /tmp/libpmcx_base.so $GRAFT_REPO_ROOT/wrf-partmc_amd/libpmcx.so; else cp $GRAFT_REPO_ROOT/alt_so/$n.so $GRAFT_REPO_ROOT/wrf-partmc_amd/libpmcx.so; fi
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/abs_$n.json 2> $GRAFT_REPO_ROOT/gpurun_out/abs_$n.err || { tail -3 $GRAFT_REPO_ROOT/gpurun_out/abs_$n.err; continue; }
  python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/abs_$n.json')); print('$n', 'bin_sort per launch %.3f'%d['device_ms']['bin_sort']['per_launch'], d['device_ms']['bin_sort']['launches'], 'wall', d['wall_ms_each_step'][1:])"
done
done
cp /tmp/libpmcx_base.so $GRAFT_REPO_ROOT/wrf-partmc_amd/libpmcx.so
