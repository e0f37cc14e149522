# TEMPORARY: builds timing-only variants of the library (PMCX_ABLATE bit mask) into ab/
set -e
cd "$(dirname "$0")/../wrf-partmc_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
for v in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DPMCX_ABLATE=$v -c pmcx_kernels.hip -o /tmp/ab_k_$v.o && \
    /opt/rocm/bin/hipcc $F -c pmcx_api.hip -o /tmp/ab_a_$v.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libpmcx_ab$v.so /tmp/ab_k_$v.o /tmp/ab_a_$v.o ) &
done
wait
ls -la ../../ab
