#!/bin/bash
# every fuzzer on the present build (outside the suite); summary lines under gpurun_out/
mkdir -p gpurun_out
: > gpurun_out/fuzz_summary.txt
for job in "fuzz_parity.py 150 11" "fuzz_transport.py 80 13" "fuzz_blocks.py 40 17" "fuzz_prep.py 80 19" "fuzz_driver.py 30 23 8" "fuzz_driver.py 30 29 8 1.5"; do
  set -- $job; name=${1%.py}
  timeout -k 10 900 python scripts/$job > gpurun_out/${name}_$3.log 2>&1
  echo "$job: exit $? : $(tail -1 gpurun_out/${name}_$3.log)" | tee -a gpurun_out/fuzz_summary.txt
done
