#!/bin/bash
# the coupled step with several weight classes / the reference's 20 species
mkdir -p gpurun_out
cd /tmp
for cfg in "c1:--steps 7" "c10:--steps 7 --n-class 10" "c10s20:--steps 7 --nx 32 --ny 32 --n-class 10 --n-spec 20" "c1s20:--steps 7 --nx 32 --ny 32 --n-spec 20" "c1small:--steps 7 --nx 32 --ny 32"; do
  n=${cfg%%:*}; f=${cfg#*:}
  timeout -k 10 400 python3 $GRAFT_REPO_ROOT/scripts/whole_step.py $f > $GRAFT_REPO_ROOT/gpurun_out/whole_$n.json 2> $GRAFT_REPO_ROOT/gpurun_out/whole_$n.err || { echo "$n failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/whole_$n.err; continue; }
  python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/whole_$n.json')); print('$n cells', d['cells'], 'wall each step', d['wall_ms_each_step'], 'events', d['coag_events']); print('   ', {k: round(v['per_launch'], 3) for k, v in d['device_ms'].items()})"
done
