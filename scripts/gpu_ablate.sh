# TEMPORARY: times each ablation variant (ab/libpmcx_abN.so) on the full domain; results invalid
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for v in "$@"; do
  PMCX_LIBRARY=$PWD/ab/libpmcx_ab$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --resort-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate=$v ms/step %.3f sampler %.3f cand/step %.3e ev/step %.3e status %d' % (d['ms_per_step'], d['kernel_ms_per_step']['coag_sample'], d['candidate_pairs_per_sec']*d['ms_per_step']/1e3, d['events_per_step'], d['status_or']))"
done
