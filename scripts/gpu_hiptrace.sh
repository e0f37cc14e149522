#!/bin/bash
# host-side view of the coupled step: HIP API statistics of scripts/whole_step.py
tag=${1:-h}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/prof_hip_$tag -- python3 $R/scripts/whole_step.py --steps 12 > $R/gpurun_out/hip_$tag.log 2>&1 || { tail -5 $R/gpurun_out/hip_$tag.log; exit 1; }
f=$(find $R/gpurun_out/prof_hip_$tag -name '*hip_api_stats.csv' | head -1)
cp "$f" $R/gpurun_out/hip_api_stats_$tag.csv
find $R/gpurun_out/prof_hip_$tag -name '*hip_api_trace.csv' -delete
head -25 $R/gpurun_out/hip_api_stats_$tag.csv
grep -o '"host_wall_ms_steady": {[^}]*}' $R/gpurun_out/hip_$tag.log
grep -o '"wall_ms_each_step": \[[^]]*\]' $R/gpurun_out/hip_$tag.log
