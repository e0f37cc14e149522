# A/B of sampler builds on ONE box: the bench line with the in-tree library, then with each
# alternative library under alt_so/ copied over it (the box's copy of the repo is scratch).
# usage: bash scripts/gpu_ab.sh <tag> [bench args...]
tag=${1:-ab}; shift
mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --resort-steps 0 "$@" > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err || { tail -5 gpurun_out/ab_${tag}_$name.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_${tag}_$name.json')); print('$name', 'ms/step %.3f' % d['ms_per_step'], 'sampler %.3f' % d['kernel_ms_per_step']['coag_sample'], 'check', d.get('oracle_check'))"
}
name=warm; run "$@"
name=base; run "$@"
cp wrf-partmc_amd/libpmcx.so /tmp/libpmcx_base.so
for f in alt_so/*.so; do
  name=$(basename $f .so); cp $f wrf-partmc_amd/libpmcx.so; run "$@"
done
cp /tmp/libpmcx_base.so wrf-partmc_amd/libpmcx.so
name=base2; run "$@"
