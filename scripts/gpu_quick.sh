# Quick check: the GPU tests (which force every sampler variant), then a full-size bench with one variant.
# usage: bash scripts/gpu_quick.sh <tag> [LANES]
tag=${1:-q}; lanes=${2:-}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/tests_$tag.log 2>&1; tail -15 gpurun_out/tests_$tag.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --sampler-lanes ${lanes:-0} > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print(json.dumps({k:d[k] for k in ["value","ms_per_step","kernel_ms_per_step","events_per_step","status_or"]}))
PY
