// gather_probe.hip -- what the chip sustains for the sampler's access shape:
// every lane of a wavefront reads W bytes at an independent random address of a
// buffer far larger than the Infinity Cache (one line per lane, no reuse).
// Gives the line rate that bounds k_coag_sample's gathers, and (under
// rocprofv3 --pmc FETCH_SIZE) the bytes the memory side moves per such line.
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_probe.hip -o gpurun_out/gather_probe
//   ./gather_probe [buffer GiB] [loads per lane]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// DEP = 1: each load's address depends on the previous value (latency chain, as a perm -> vtot
// lookup); DEP = 0: ILP independent loads in flight per lane.
template <int W, int ILP, int DEP>
__global__ __launch_bounds__(256) void k_gather(const uint64_t *buf, uint64_t n_lines, int line_b,
                                                int iters, uint64_t *out)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t acc = 0, h = mix(tid + 1) | 1;
    for (int it = 0; it < iters; it++) {
        uint64_t v[ILP][W / 8];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            // cheap address arithmetic (the probe must not be ALU-bound): xorshift32, n_lines = 2^k
            uint32_t x = (uint32_t)h + (DEP ? (uint32_t)acc : 0u);
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            h = x;
            const uint64_t line = ((uint64_t)x * 2654435761ull >> 8) & (n_lines - 1);
            const uint64_t *p = buf + line * (uint64_t)(line_b / 8);
#pragma unroll
            for (int j = 0; j < W / 8; j++) v[k][j] = __builtin_nontemporal_load(p + j);
        }
#pragma unroll
        for (int k = 0; k < ILP; k++)
#pragma unroll
            for (int j = 0; j < W / 8; j++) acc += v[k][j];
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int W, int ILP, int DEP>
static void run(const char *name, const uint64_t *buf, uint64_t bytes, int line_b, int iters, uint64_t *out)
{
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint64_t n_lines = 1;
    while (n_lines * 2 * line_b <= bytes) n_lines *= 2; // largest power of two that fits
    k_gather<W, ILP, DEP><<<blocks, threads>>>(buf, n_lines, line_b, 2, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_gather<W, ILP, DEP><<<blocks, threads>>>(buf, n_lines, line_b, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * threads * iters * ILP;
    printf("%-34s line=%3d B  %8.2f ms  %7.2f G loads/s  useful %6.1f GB/s  at 64 B/line %7.1f GB/s  at 128 B/line %7.1f GB/s\n",
           name, line_b, ms, loads / ms * 1e-6, loads * W / ms * 1e-6, loads * 64 / ms * 1e-6,
           loads * 128 / ms * 1e-6);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 48.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~(uint64_t)4095;
    uint64_t *buf, *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    printf("buffer %.1f GiB, %d iterations, 2048 workgroups x 256 lanes\n", gib, iters);
    run<8, 1, 1>("8 B, dependent chain", buf, bytes, 128, iters, out);
    run<8, 1, 0>("8 B, 1 in flight", buf, bytes, 128, iters, out);
    run<8, 4, 0>("8 B, 4 in flight", buf, bytes, 128, iters, out);
    run<8, 8, 0>("8 B, 8 in flight", buf, bytes, 128, iters, out);
    run<8, 8, 0>("8 B, 8 in flight (64-B stride)", buf, bytes, 64, iters, out);
    run<16, 8, 0>("16 B, 8 in flight", buf, bytes, 128, iters, out);
    run<64, 4, 0>("64 B per lane, 4 in flight", buf, bytes, 64, iters, out);
    run<64, 4, 0>("64 B per lane, 4 in flight", buf, bytes, 128, iters, out);
    run<128, 2, 0>("128 B per lane, 2 in flight", buf, bytes, 128, iters, out);
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
