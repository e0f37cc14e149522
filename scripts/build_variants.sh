# TEMPORARY: builds tuning variants of the library into ab/ : args "name:flags" ...
set -e
cd "$(dirname "$0")/../wrf-partmc_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
mkdir -p ../../ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $F $flags -c pmcx_kernels.hip -o /tmp/v_k_$name.o && \
    /opt/rocm/bin/hipcc $F -c pmcx_api.hip -o /tmp/v_a_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libpmcx_$name.so /tmp/v_k_$name.o /tmp/v_a_$name.o ) &
done
wait
