#!/bin/bash
mkdir -p gpurun_out
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
for f in alt_so/cap16k.so alt_so/m3000.so; do
  cp $f wrf-partmc_amd/libpmcx.so
  echo "== $f"
  timeout -k 10 600 python -m pytest tests/test_gpu_classes.py -q -x -k "many_classes" 2>&1 | tail -4
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
