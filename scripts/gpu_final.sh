#!/bin/bash
# what the driver runs at round end: the -m gpu suite, smoke(), the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/final_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/final_tests.log
tail -4 gpurun_out/final_tests.log
grep -q "tests exit 0" gpurun_out/final_tests.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 1500 gpurun_out/final_bench.json
