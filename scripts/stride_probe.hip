// stride_probe.hip -- what a streaming pass pays for reading one 8-byte field of interleaved
// records: (a) field of 16-byte records (every other double), (b) records blocked four at a
// time, x*4 | y*4 per 64-byte line (the first 32-byte sector of every line), (c) a dense plane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k_read(const double *buf, int64_t n, double *out)
{
    double acc = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = MODE == 0 ? 2 * i : MODE == 1 ? ((i >> 2) * 8 + (i & 3)) : i;
        acc += buf[j];
    }
    if (acc == 1.2345e300) out[0] = acc;
}
template <int MODE>
static int run(const char *name, const double *buf, int64_t n, double *out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_read<MODE><<<256 * 32, 256>>>(buf, n / 8, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); k_read<MODE><<<256 * 32, 256>>>(buf, n, out); CK(hipEventRecord(b));
    CK(hipEventSynchronize(b)); float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-46s %7.3f ms  %6.1f G elements/s  (%.0f GB/s of 8-byte payload)\n", name, ms, n / ms * 1e-6, n * 8.0 / ms * 1e-6);
    return 0;
}
int main()
{
    const int64_t n = 1ll << 30; // elements read
    double *buf, *out; CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, n * 16));
    if (run<0>("field of 16-byte records (stride 16 B)", buf, n, out)) return 1;
    if (run<1>("blocked x*4|y*4 per 64-byte line (sector 0)", buf, n, out)) return 1;
    if (run<2>("dense plane", buf, n, out)) return 1;
    return 0;
}
