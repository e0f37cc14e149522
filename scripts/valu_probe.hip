// valu_probe.hip -- issue cost of the vector instructions the pair sampler is made of, on gfx950.
// Each test runs N_IT iterations of 8 independent chains of one instruction per lane and reports
// shader cycles per wave-instruction, for one wavefront alone on its SIMD (latency-free issue
// rate) and for three wavefronts sharing it (what the sampler runs with).
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe scripts/valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int N_IT = 16384;

template <int OP> __device__ __forceinline__ void step(double (&d)[8], uint32_t (&u)[8], float (&f)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (OP == 0) d[k] = __builtin_fma(d[k], 1.0000001, 1e-9);                 // v_fma_f64
        if (OP == 1) d[k] = d[k] * 1.0000001;                                     // v_mul_f64
        if (OP == 2) d[k] = d[k] + 1e-9;                                          // v_add_f64
        if (OP == 3) u[k] = u[k] * 0xD2511F53u;                                   // v_mul_lo_u32
        if (OP == 4) u[k] = __umulhi(u[k], 0xD2511F53u) ^ (uint32_t)k;            // v_mul_hi_u32
        if (OP == 5) { const uint64_t p = (uint64_t)u[k] * 0xCD9E8D57u; u[k] = (uint32_t)(p >> 32) ^ (uint32_t)p; } // mad_u64_u32 / lo+hi
        if (OP == 6) f[k] = __builtin_fmaf(f[k], 1.0000001f, 1e-9f);              // v_fma_f32
        if (OP == 7) f[k] = __builtin_amdgcn_rcpf(f[k]) + 1.5f;                   // v_rcp_f32
        if (OP == 8) f[k] = __builtin_amdgcn_exp2f(f[k]) * 0.25f;                 // v_exp_f32
        if (OP == 9) f[k] = __builtin_amdgcn_logf(f[k]) + 3.0f;                   // v_log_f32
        if (OP == 10) f[k] = __builtin_amdgcn_sqrtf(f[k]) + 1.0f;                 // v_sqrt_f32
        if (OP == 11) d[k] = __builtin_amdgcn_rcp(d[k]) + 1.5;                    // v_rcp_f64
        if (OP == 12) d[k] = __builtin_amdgcn_sqrt(d[k]) + 1.0;                   // v_sqrt_f64
        if (OP == 13) d[k] = d[k] / 1.0000001;                                    // IEEE f64 division
        if (OP == 14) d[k] = __builtin_sqrt(d[k]) + 1.0;                          // IEEE f64 sqrt
        if (OP == 15) u[k] = (u[k] ^ (u[k] >> 7)) + 0x9E3779B9u;                  // 32-bit xor/shift/add
        if (OP == 16) d[k] = (double)u[k] + d[k];                                 // v_cvt_f64_u32 + add
        if (OP == 17) f[k] = (float)d[k] + f[k];                                  // v_cvt_f32_f64 + add
    }
}

template <int OP> __global__ void k_probe(uint64_t *cycles, double *sink)
{
    double d[8]; uint32_t u[8]; float f[8];
    for (int k = 0; k < 8; k++) { d[k] = 1.0 + 1e-3 * (threadIdx.x + k); u[k] = 12345u * (threadIdx.x + 7 * k + 1); f[k] = 1.5f + 0.01f * k; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; it++) step<OP>(d, u, f);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 8; k++) s += d[k] + u[k] + f[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> static void run(const char *name, int per_instr)
{
    uint64_t *c; double *sink;
    hipMalloc(&c, 8 * 4096); hipMalloc(&sink, 8 * 64 * 4096);
    for (int waves_per_simd : {1, 3}) {
        // one workgroup per CU, 4 * waves_per_simd wavefronts: the CU's four SIMDs get that many each
        const int block = 64 * 4 * waves_per_simd, grid = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_probe<OP>, dim3(grid), dim3(block), 0, 0, c, sink); // warm-up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<OP>, dim3(grid), dim3(block), 0, 0, c, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(grid * 4 * waves_per_simd);
        hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (uint64_t x : h) mean += (double)x;
        mean /= h.size();
        // s_memtime counts at 100 MHz on this chip; the shader clock is read from the device
        int khz = 0;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
        const double cyc = mean * (khz / 1e5);     // 100 MHz ticks -> shader cycles
        const double n_instr = (double)N_IT * 8 * per_instr;
        printf("%-30s %d wave(s)/SIMD: %7.2f cycles per wave-instruction by s_memtime, %7.2f by the kernel's time "
               "(per SIMD: %.2f)\n", name, waves_per_simd, cyc / n_instr, ms * 1e-3 * khz * 1e3 / n_instr,
               ms * 1e-3 * khz * 1e3 / n_instr / waves_per_simd);
    }
    hipFree(c); hipFree(sink);
}

int main()
{
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("shader clock %d kHz; s_memtime assumed 100 MHz\n", khz);
    run<0>("v_fma_f64", 1); run<1>("v_mul_f64", 1); run<2>("v_add_f64", 1);
    run<3>("v_mul_lo_u32", 1); run<4>("v_mul_hi_u32 (+xor)", 2); run<5>("u32 x u32 -> u64 (hi ^ lo)", 2);
    run<6>("v_fma_f32", 1); run<7>("v_rcp_f32 (+add)", 2); run<8>("v_exp_f32 (+mul)", 2);
    run<9>("v_log_f32 (+add)", 2); run<10>("v_sqrt_f32 (+add)", 2); run<11>("v_rcp_f64 (+add)", 2);
    run<12>("v_sqrt_f64 (+add)", 2); run<13>("f64 division (IEEE)", 1); run<14>("f64 sqrt (IEEE) (+add)", 1);
    run<15>("u32 xor/shift/add x3", 3); run<16>("cvt_f64_u32 + add", 2); run<17>("cvt_f32_f64 + add", 2);
    return 0;
}
