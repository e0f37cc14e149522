#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_species20.py tests/test_gpu_parity.py tests/test_gpu_components.py -q -x -k "dispatch or species or diag" > gpurun_out/r4s.log 2>&1; echo "exit $?" >> gpurun_out/r4s.log
tail -4 gpurun_out/r4s.log
grep -q "exit 0" gpurun_out/r4s.log || exit 1
cd /tmp
for cfg in "c1s20:--steps 4 --nx 32 --ny 32 --n-spec 20" "c10s20:--steps 4 --nx 32 --ny 32 --n-class 10 --n-spec 20"; do
  n=${cfg%%:*}; f=${cfg#*:}
  timeout -k 10 400 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py $f > $GRAFT_REPO_ROOT/gpurun_out/whole_$n.json 2> $GRAFT_REPO_ROOT/gpurun_out/whole_$n.err || { echo "$n failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/whole_$n.err; continue; }
  python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/whole_$n.json')); print('$n', {k: round(v['per_launch'], 3) for k, v in d['device_ms'].items() if k in ('diagnostics','transport','bin_sort')})"
done
