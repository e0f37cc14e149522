# TEMPORARY: times each tuning variant (ab/libpmcx_NAME.so) on the full domain and a rank's share at N = 8
for v in "$@"; do
  for ny in 100 13; do
  PMCX_SAMPLER_LANES=1 PMCX_LIBRARY=$PWD/ab/libpmcx_$v.so timeout -k 10 120 python bench.py --ny $ny --no-cpu-baseline --steps 4 --warmup 1 --resort-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v ny=$ny ms/step %.3f sampler %.3f ev/step %.3e status %d' % (d['ms_per_step'], d['kernel_ms_per_step']['coag_sample'], d['events_per_step'], d['status_or']))"
  done
done
