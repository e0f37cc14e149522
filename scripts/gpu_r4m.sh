#!/bin/bash
mkdir -p gpurun_out
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
for n in dbg512 dbg16k; do
  cp alt_so/$n.so wrf-partmc_amd/libpmcx.so
  echo "== $n"
  timeout -k 10 300 python scripts/debug_listed.py $n 2>&1 | tail -12
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
