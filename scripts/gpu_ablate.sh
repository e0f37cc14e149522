# Timing ablations of the lane-per-cell sampler (PMCX_DBG bits; results are invalid, only the
# sampler time is read): 1 Poisson, 2 draw Philox, 4 kernel evaluation, 16 particle loads
for d in 0 1 2 4 16 7 23; do
  PMCX_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --resort-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$d', round(d['kernel_ms_per_step']['coag_sample'],3), 'ms  events/step', int(d['events_per_step']), 'cand/s %.3g' % d['candidate_pairs_per_sec'])"
done
