# A/B of sampler builds on ONE box: the bench line with the in-tree library and with each
# alternative library under alt_so/ copied over it (the box's copy of the repo is scratch),
# interleaved REPS times after a warm-up run; the minimum and the median of each are printed
# (boxes drift by several per cent within a minute: single runs do not separate 2 % effects).
# usage: [REPS=3] bash scripts/gpu_ab.sh <tag> [bench args...]
tag=${1:-ab}; shift
reps=${REPS:-3}
mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --resort-steps 0 "$@" > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err || { tail -5 gpurun_out/ab_${tag}_$name.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_${tag}_$name.json')); print('$name', '%.3f' % d['kernel_ms_per_step']['coag_sample'])" >> gpurun_out/ab_${tag}.txt
}
: > gpurun_out/ab_${tag}.txt
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
name=warm; run "$@"
for r in $(seq $reps); do
  cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so; name=base; run "$@"
  for f in $(ls alt_so/*.so 2>/dev/null); do
    name=$(basename $f .so); cp $f wrf-partmc_amd/libpmcx.so; run "$@"
  done
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
python3 - <<PY
import collections, statistics
d=collections.OrderedDict()
for l in open('gpurun_out/ab_${tag}.txt'):
    n,v=l.split(); d.setdefault(n,[]).append(float(v))
for n,v in d.items():
    print('%-40s min %.3f  median %.3f  n=%d  %s' % (n, min(v), statistics.median(v), len(v), ' '.join('%.3f'%x for x in v)))
PY
