// pmcx_kernels.hip -- gfx950 kernels of the PartMC per-cell microphysics path.
//
// Path replaced (one Fortran loop body per grid cell, serial per MPI rank):
//   partmc_timestep                 interface/wrf_pmc_driver.F90:169-254
//     mc_coag(COAG_KERNEL_TYPE_BROWN, ...)               driver:192-194
//     aero_state_rebalance                               driver:248-250
//   partmc_process observables                           driver:783,838-857
//
// Layout in HBM (structure of arrays, one slab of `cap` slots per cell):
//   vol   [n_cells][n_spec][cap] f64   particle%vol(:)
//   vtot  [n_cells][cap] f64           cached sum_s vol
//   mass  [n_cells][cap] f64           cached sum_s density_s*vol
//   wclass[n_cells][cap] i32, id [..] i64, lct/gct [..] f64,
//   n_orig[n_cells][n_source][cap] i32
// Kernels:
//   k_bin_sort     one 256-thread workgroup per cell; stable counting sort of
//                  the slots by (class, bin) -> perm, gcnt, goff.  Streams
//                  vtot once: the HBM-bound pass (12 B per particle).
//   k_coag_sample  one 64-lane wavefront per cell; candidate pairs of the
//                  sequential PartMC sweep evaluated 64 at a time, first
//                  accepted pair found by ballot, applied, later slots redrawn.
//   k_coag_apply   one wavefront per cell; merges species volumes along the
//                  event log, refreshes caches, closes holes (bitmap +
//                  wave prefix sums).
//   k_rebalance, k_diagnostics, k_kernel_bounds, k_generate: full-population
//                  or per-cell table passes around the above.
// All arithmetic that feeds a decision is fp64 with contraction off.
#include "pmcx_kernels.h"
#include "pmcx_device_math.h"

namespace pmcx {

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double ld_f64(const double *p)
{
    const uint64_t u = __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return u2d(u);
}
__device__ __forceinline__ void st_f64(double *p, double v)
{
    __hip_atomic_store((uint64_t *)p, d2u(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u32(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// Exact bin of a volume on the log-spaced grid: a bit-level estimate of
// log2(v) lands within one bin of the answer and two edge compares settle it.
__device__ __forceinline__ int bin_of_fast(const double *v_edge, int n_bin, double v,
                                           double log2_v0, double inv_dlog2)
{
    const uint64_t u = d2u(v);
    const double l2 = (double)((int)((u >> 52) & 0x7FF) - 1023) +
                      (double)(u & 0x000FFFFFFFFFFFFFull) * 0x1.0p-52;
    int b = (int)((l2 - log2_v0) * inv_dlog2);
    b = b < 0 ? 0 : (b > n_bin - 1 ? n_bin - 1 : b);
    // settle (at most a couple of steps; loops only guard pathological input)
    while (b > 0 && v < v_edge[b]) b--;
    while (b < n_bin - 1 && v >= v_edge[b + 1]) b++;
    return b;
}

// ------------------------------------------------------------ upload/download

// packed [n_total] (cell-major) -> slab [n_cells][cap]
__global__ void k_scatter_f64(const double *__restrict__ src, double *__restrict__ dst,
                              const int64_t *__restrict__ part_offset, int n_up, int cap,
                              int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride)
{
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[(int64_t)k * dst_cell_stride + r * dst_row_stride + i] =
                src[r * src_row_stride + o0 + i];
}
__global__ void k_scatter_i32(const int32_t *__restrict__ src, int32_t *__restrict__ dst,
                              const int64_t *__restrict__ part_offset, int n_up, int cap,
                              int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride)
{
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[(int64_t)k * dst_cell_stride + r * dst_row_stride + i] =
                src[r * src_row_stride + o0 + i];
}
__global__ void k_gather_f64(double *__restrict__ dst, const double *__restrict__ src,
                             const int64_t *__restrict__ part_offset, int64_t src_cell_stride,
                             int64_t dst_row_stride, int n_rows, int64_t src_row_stride)
{
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[r * dst_row_stride + o0 + i] =
                src[(int64_t)k * src_cell_stride + r * src_row_stride + i];
}
__global__ void k_gather_i32(int32_t *__restrict__ dst, const int32_t *__restrict__ src,
                             const int64_t *__restrict__ part_offset, int64_t src_cell_stride,
                             int64_t dst_row_stride, int n_rows, int64_t src_row_stride)
{
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[r * dst_row_stride + o0 + i] =
                src[(int64_t)k * src_cell_stride + r * src_row_stride + i];
}
__global__ void k_set_n_part(int32_t *n_part, const int64_t *part_offset, int n_up)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_up) n_part[k] = (int32_t)(part_offset[k + 1] - part_offset[k]);
}

// vtot/mass from the species volumes (sequential sum over species)
__global__ void k_refresh_caches(DeviceState st, int cell_first, int n_cells_do)
{
    const int c = cell_first + blockIdx.y;
    const int n = st.n_part[c];
    const int64_t cb = (int64_t)c * st.cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double vt = 0.0, ms = 0.0;
        for (int s = 0; s < st.n_spec; s++) {
            const double v = st.vol[(cb * st.n_spec) + (int64_t)s * st.cap + i];
            vt = vt + v;
            ms = ms + st.density[s] * v;
        }
        st.vtot[cb + i] = vt;
        st.mass[cb + i] = ms;
    }
}

// ------------------------------------------------------------ kernel bounds
//
// est_k_minmax_binned_unweighted [upstream]: per cell, K_max(b1,b2) = max of K
// over 3x3 sample volumes of the two bins and (rho_min|rho_max)^2.

__global__ __launch_bounds__(256) void k_kernel_bounds(DeviceState st, int ktype)
{
    extern __shared__ __align__(16) unsigned char smem[];
    PartProps *pp = (PartProps *)smem; // [n_bin*6]
    const int c = blockIdx.x;
    const int B = st.n_bin;
    const EnvDerived env = env_derive(st.temp[c], st.pressure[c]);
    const KernelParams kp{st.const_k, st.additive_beta};
    const bool props = kernel_needs_props(ktype);
    for (int j = threadIdx.x; j < B * 6; j += blockDim.x) {
        const int d = j & 1, bs = j >> 1; // bs = b*3+s
        const double v = st.v_samp[bs];
        const double rho = d ? st.rho_max : st.rho_min;
        PartProps q{0, 0, 0, 0, 0};
        if (props) q = particle_props(v, v * rho, env);
        pp[j] = q;
    }
    __syncthreads();
    double *km = st.kmax + (int64_t)c * B * B;
    for (int j = threadIdx.x; j < B * B; j += blockDim.x) {
        const int b1 = j / B, b2 = j - b1 * B;
        double m = 0.0;
        for (int s1 = 0; s1 < 3; s1++)
            for (int s2 = 0; s2 < 3; s2++)
                for (int d1 = 0; d1 < 2; d1++)
                    for (int d2 = 0; d2 < 2; d2++) {
                        const double k = kernel_pair(
                            ktype, st.v_samp[3 * b1 + s1], st.v_samp[3 * b2 + s2],
                            pp[(b1 * 3 + s1) * 2 + d1], pp[(b2 * 3 + s2) * 2 + d2], kp);
                        if (k > m) m = k;
                    }
        km[j] = m;
    }
}

// ---------------------------------------------------------------- bin sort
//
// aero_state_sort [upstream; called at wrf_pmc_init.F90:1279 and inside
// mc_coag]: groups g = class*n_bin + bin, ascending slot order inside a
// group.  Stable counting sort with per-thread contiguous ownership of
// positions, so the order never depends on atomics or scheduling.  With more
// than one class it runs as two stable passes (bin, then class).

struct SortLds {
    uint16_t *tab;   // [D][256] per-digit per-thread counts -> write cursors
    uint32_t *dtot;  // [D] digit totals, then exclusive bases
    uint8_t *key;    // [n] bin of slot
    uint8_t *cls;    // [n] class of slot (n_class > 1)
    uint16_t *ord;   // [n] order after the bin pass (n_class > 1)
    uint32_t *gcnt;  // [G] (n_class > 1)
    double *edge;    // [n_bin+1]
};

// tab[d][t] += 1 for every position owned by thread t; then per-digit
// exclusive scan over t, digit totals, exclusive scan over digits.
template <typename DigitOf>
__device__ __forceinline__ void count_and_scan(const SortLds &L, int D, int n, int chunk,
                                               DigitOf digit_of)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int j = t; j < D * 256; j += 256) L.tab[j] = 0;
    __syncthreads();
    {
        const int j0 = t * chunk, j1 = min(n, j0 + chunk);
        for (int j = j0; j < j1; j++) L.tab[digit_of(j) * 256 + t] += 1;
    }
    __syncthreads();
    for (int d = wave; d < D; d += 4) {
        uint16_t *row = L.tab + d * 256;
        const uint32_t a0 = row[4 * lane], a1 = row[4 * lane + 1], a2 = row[4 * lane + 2],
                       a3 = row[4 * lane + 3];
        const uint32_t s = a0 + a1 + a2 + a3;
        const uint32_t inc = wave_incl_scan_u32(s, lane);
        const uint32_t ex = inc - s;
        row[4 * lane] = (uint16_t)ex;
        row[4 * lane + 1] = (uint16_t)(ex + a0);
        row[4 * lane + 2] = (uint16_t)(ex + a0 + a1);
        row[4 * lane + 3] = (uint16_t)(ex + a0 + a1 + a2);
        if (lane == 63) L.dtot[d] = inc;
    }
    __syncthreads();
}

// exclusive scan in place over an LDS u32 array of length n (n <= 256*K)
__device__ __forceinline__ void block_excl_scan_u32(uint32_t *a, int n, uint32_t *wsum)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + 255) / 256;
    const int j0 = t * per, j1 = min(n, j0 + per);
    uint32_t s = 0;
    for (int j = j0; j < j1; j++) s += a[j];
    const uint32_t inc = wave_incl_scan_u32(s, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - s;
    for (int w = 0; w < wave; w++) base += wsum[w];
    for (int j = j0; j < j1; j++) {
        const uint32_t x = a[j];
        a[j] = base;
        base += x;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bin_sort(DeviceState st)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wsum[4];
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    const int B = st.n_bin, C = st.n_class, G = B * C;
    const int n = st.n_part[c];
    const int cap = st.cap;
    const int64_t cb = (int64_t)c * cap;
    const int D = B > C ? B : C;

    SortLds L;
    size_t o = 0;
    L.edge = (double *)(smem + o); o += sizeof(double) * (size_t)(B + 1);
    L.dtot = (uint32_t *)(smem + o); o += sizeof(uint32_t) * (size_t)D;
    L.gcnt = (uint32_t *)(smem + o); o += (C > 1) ? sizeof(uint32_t) * (size_t)G : 0;
    L.tab = (uint16_t *)(smem + o); o += sizeof(uint16_t) * 256 * (size_t)D;
    L.ord = (uint16_t *)(smem + o); o += (C > 1) ? sizeof(uint16_t) * (size_t)cap : 0;
    L.key = (uint8_t *)(smem + o); o += (size_t)cap;
    L.cls = (uint8_t *)(smem + o);

    for (int j = t; j <= B; j += 256) L.edge[j] = st.v_edge[j];
    if (C > 1) for (int j = t; j < G; j += 256) L.gcnt[j] = 0;
    __syncthreads();

    // keys: one coalesced streaming read of vtot (8 B per particle)
    const double *vt = st.vtot + cb;
    const int32_t *wc = st.wclass + cb;
    const double v_lo = L.edge[0], v_hi = L.edge[B];
    int oog = 0;
    for (int i = t; i < n; i += 256) {
        const double v = vt[i];
        oog |= (v < v_lo) | (v >= v_hi);
        const int b = bin_of_fast(L.edge, B, v, st.log2_v0, st.inv_dlog2);
        L.key[i] = (uint8_t)b;
        if (C > 1) {
            const int cl = wc[i];
            L.cls[i] = (uint8_t)cl;
            atomicAdd(&L.gcnt[cl * B + b], 1u);
        }
    }
    if (oog) atomicOr(&st.status[c], ST_OUT_OF_GRID);
    __syncthreads();

    const int chunk = (n + 255) / 256;
    uint32_t *perm = st.perm + cb;
    int32_t *gcnt_g = st.gcnt + (int64_t)c * G;
    int32_t *goff_g = st.goff + (int64_t)c * G;

    // pass 1: by bin
    count_and_scan(L, B, n, chunk, [&](int j) { return (int)L.key[j]; });
    if (C == 1) {
        for (int j = t; j < B; j += 256) gcnt_g[j] = (int32_t)L.dtot[j];
    }
    block_excl_scan_u32(L.dtot, B, wsum);
    if (C == 1) {
        for (int j = t; j < B; j += 256) goff_g[j] = (int32_t)L.dtot[j];
    }
    {
        const int j0 = t * chunk, j1 = min(n, j0 + chunk);
        for (int j = j0; j < j1; j++) {
            const int d = L.key[j];
            const uint32_t pos = L.dtot[d] + L.tab[d * 256 + t];
            L.tab[d * 256 + t] += 1;
            if (C == 1) perm[pos] = (uint32_t)j; else L.ord[pos] = (uint16_t)j;
        }
    }
    if (C == 1) return;
    __syncthreads();

    // pass 2: by class, over the bin-sorted order
    count_and_scan(L, C, n, chunk, [&](int j) { return (int)L.cls[L.ord[j]]; });
    block_excl_scan_u32(L.dtot, C, wsum);
    {
        const int j0 = t * chunk, j1 = min(n, j0 + chunk);
        for (int j = j0; j < j1; j++) {
            const int slot = L.ord[j];
            const int d = L.cls[slot];
            const uint32_t pos = L.dtot[d] + L.tab[d * 256 + t];
            L.tab[d * 256 + t] += 1;
            perm[pos] = (uint32_t)slot;
        }
    }
    for (int j = t; j < G; j += 256) gcnt_g[j] = (int32_t)L.gcnt[j];
    __syncthreads();
    block_excl_scan_u32(L.gcnt, G, wsum);
    for (int j = t; j < G; j += 256) goff_g[j] = (int32_t)L.gcnt[j];
}

// ------------------------------------------------------------- coag sampler
//
// mc_coag sweep of one cell by one wavefront.  Sequential semantics of the
// reference loop nest (classes, bins, samples in order; every accepted pair
// changes the lists the next sample draws from) are kept exactly: the wave
// evaluates up to 64 upcoming candidates of the current row against the
// current lists, the first accepted one (ballot + ctz) is applied, and the
// Poisson counts of the row's later slots are redrawn from the new counts.
// Draws are addressed by (seed, cell, step, slot, sample), so a redraw with
// unchanged counts reproduces the same number.

struct CellLds {
    uint16_t *cnt;      // [G] current list length
    uint16_t *nb;       // [G] entries still in the sorted base segment
    uint16_t *off;      // [G] base segment offset in perm
    uint16_t *last_app; // [G] newest appended entry of the group (0xFFFF none)
    uint16_t *app_slot; // [max_appends]
    uint16_t *app_prev; // [max_appends]
    uint32_t *ns;       // [n_bin] Poisson counts of the current row
    uint64_t *P;        // [n_bin] inclusive prefix of remaining samples
    uint64_t *mask;     // [n_class][2] non-empty bins bitmask
};

__host__ __device__ inline size_t sampler_lds_bytes(int G, int B, int C, int max_appends)
{
    size_t o = 0;
    o += sizeof(uint64_t) * (size_t)B;      // P
    o += sizeof(uint64_t) * 2 * (size_t)C;  // mask
    o += sizeof(uint32_t) * (size_t)B;      // ns
    o += sizeof(uint16_t) * 4 * (size_t)G;  // cnt nb off last_app
    o += sizeof(uint16_t) * 2 * (size_t)max_appends;
    return (o + 15) & ~(size_t)15;
}

struct CellCtx {
    CellLds L;
    uint32_t *perm;
    double *vtot, *mass;
};

__device__ __forceinline__ uint32_t list_get(const CellCtx &X, int g, int i)
{
    const int nb = X.L.nb[g];
    if (i < nb) return ld_u32(&X.perm[X.L.off[g] + i]);
    int e = X.L.last_app[g];
    for (int k = X.L.cnt[g] - 1 - i; k > 0; k--) e = X.L.app_prev[e];
    return X.L.app_slot[e];
}

// uniform across the wave: every lane performs the same LDS writes
__device__ __forceinline__ void list_set(const CellCtx &X, int g, int i, uint32_t val, int lane)
{
    const int nb = X.L.nb[g];
    if (i < nb) {
        if (lane == 0) st_u32(&X.perm[X.L.off[g] + i], val);
    } else {
        int e = X.L.last_app[g];
        for (int k = X.L.cnt[g] - 1 - i; k > 0; k--) e = X.L.app_prev[e];
        X.L.app_slot[e] = (uint16_t)val;
    }
}

// swap-with-last removal of logical position i; returns the old last index
__device__ __forceinline__ int list_remove(const CellCtx &X, int g, int i, int B, int lane)
{
    const int last = (int)X.L.cnt[g] - 1;
    if (i != last) list_set(X, g, i, list_get(X, g, last), lane);
    if (last >= (int)X.L.nb[g]) X.L.last_app[g] = X.L.app_prev[X.L.last_app[g]];
    else X.L.nb[g] = (uint16_t)last;
    X.L.cnt[g] = (uint16_t)last;
    if (last == 0) {
        const int cl = g / B, b = g - cl * B;
        X.L.mask[cl * 2 + (b >> 6)] &= ~(1ull << (b & 63));
    }
    return last;
}

__device__ __forceinline__ void list_append(const CellCtx &X, int g, uint32_t slot, int e, int B)
{
    X.L.app_slot[e] = (uint16_t)slot;
    X.L.app_prev[e] = X.L.last_app[g];
    X.L.last_app[g] = (uint16_t)e;
    const int n = X.L.cnt[g];
    X.L.cnt[g] = (uint16_t)(n + 1);
    if (n == 0) {
        const int cl = g / B, b = g - cl * B;
        X.L.mask[cl * 2 + (b >> 6)] |= (1ull << (b & 63));
    }
}

__device__ __forceinline__ int next_bit(const uint64_t *m2, int from)
{
    if (from < 64) {
        const uint64_t w = m2[0] & (~0ull << from);
        if (w) return __builtin_ctzll(w);
        from = 64;
    }
    if (from < 128) {
        const uint64_t w = m2[1] & (~0ull << (from - 64));
        if (w) return 64 + __builtin_ctzll(w);
    }
    return -1;
}

__global__ __launch_bounds__(256) void k_coag_sample(DeviceState st, int ktype, double del_t,
                                                     uint64_t seed, uint32_t step,
                                                     size_t lds_per_cell)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int c = blockIdx.x * wpb + wave;
    if (c >= st.n_cells) return; // no workgroup barrier is used below
    const int B = st.n_bin, C = st.n_class, G = B * C;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = st.cell_id_base + (uint32_t)c;

    CellCtx X;
    {
        unsigned char *base = smem + lds_per_cell * wave;
        size_t o = 0;
        X.L.P = (uint64_t *)(base + o); o += sizeof(uint64_t) * (size_t)B;
        X.L.mask = (uint64_t *)(base + o); o += sizeof(uint64_t) * 2 * (size_t)C;
        X.L.ns = (uint32_t *)(base + o); o += sizeof(uint32_t) * (size_t)B;
        X.L.cnt = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.nb = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.off = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.last_app = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.app_slot = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)st.max_appends;
        X.L.app_prev = (uint16_t *)(base + o);
    }
    X.perm = st.perm + cb;
    X.vtot = st.vtot + cb;
    X.mass = st.mass + cb;

    const int32_t *gcnt_g = st.gcnt + (int64_t)c * G;
    const int32_t *goff_g = st.goff + (int64_t)c * G;
    for (int g = lane; g < G; g += 64) {
        const int n = gcnt_g[g];
        X.L.cnt[g] = (uint16_t)n;
        X.L.nb[g] = (uint16_t)n;
        X.L.off[g] = (uint16_t)goff_g[g];
        X.L.last_app[g] = 0xFFFF;
    }
    for (int cl = 0; cl < C; cl++) {
        const uint64_t m0 = __ballot(lane < B && gcnt_g[cl * B + lane] > 0);
        const uint64_t m1 = __ballot(64 + lane < B && gcnt_g[cl * B + (64 + lane < B ? 64 + lane : 0)] > 0);
        X.L.mask[cl * 2] = m0;
        X.L.mask[cl * 2 + 1] = m1;
    }
    wave_sync();

    const EnvDerived env = env_derive(st.temp[c], st.pressure[c]);
    const KernelParams kp{st.const_k, st.additive_beta};
    const bool use_props = kernel_needs_props(ktype);
    const double *kmax = st.kmax + (int64_t)c * B * B;
    const double *mag = st.mag + (int64_t)c * C;
    int4 *evlog = st.events + (int64_t)c * st.max_events;

    int n_ev = 0, n_app = 0, tot_coag = 0, status = 0;
    uint64_t tot_samp = 0;
    bool stop = false;

    for (int c1 = 0; c1 < C && !stop; c1++)
    for (int c2 = 0; c2 <= c1 && !stop; c2++) {
        const double n1 = mag[c1], n2 = mag[c2];
        const int cc = (n1 < n2) ? c1 : c2;
        const double nmin = (n1 < n2) ? n1 : n2;
        const double fmax = (n1 < n2) ? n2 : n1;
        const double factor = n1 * n2 / nmin;
        int b1 = -1;
        while (!stop) {
            b1 = next_bit(&X.L.mask[c1 * 2], b1 + 1);
            if (b1 < 0 || b1 >= B) break;
            const int gA = c1 * B + b1;
            const int b2s = (c1 == c2) ? b1 : 0;

            // Poisson counts of slots (gA, (c2,b2)), b2 >= from
            auto compute_ns = [&](int from) {
                const int nA = X.L.cnt[gA];
                for (int k = 0; k < 2; k++) {
                    const int b2 = k * 64 + lane;
                    if (b2 < from || b2 >= B) continue;
                    const int gB = c2 * B + b2;
                    const int nB = X.L.cnt[gB];
                    const bool same = (gA == gB);
                    uint32_t ns = 0;
                    if (nA > 0 && nB > 0 && !(same && nA < 2)) {
                        const double k_max = kmax[b1 * B + b2] * fmax;
                        const double n_possible =
                            same ? (double)nA * ((double)nA - 1.0) / 2.0
                                 : (double)nA * (double)nB;
                        const double mean = k_max * del_t * n_possible;
                        uint32_t w[4];
                        rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB,
                                  SITE_POISSON, 0, w);
                        ns = (uint32_t)poisson(mean, u53(w[0], w[1]));
                    }
                    X.L.ns[b2] = ns;
                }
                wave_sync();
            };
            auto sum_ns = [&](int from, int to) { // sum ns[from..to)
                uint64_t s = 0;
                for (int k = 0; k < 2; k++) {
                    const int b2 = k * 64 + lane;
                    if (b2 >= from && b2 < to) s += X.L.ns[b2];
                }
                return wave_sum_u64(s);
            };

            compute_ns(b2s);
            int counted = b2s, cur_b2 = b2s;
            uint32_t cur_s0 = 0;

            while (!stop) {
                // inclusive prefix of remaining samples over b2 in [cur_b2, B)
                uint64_t T;
                {
                    uint64_t r0 = 0, r1 = 0;
                    if (lane >= cur_b2 && lane < B)
                        r0 = X.L.ns[lane] - (lane == cur_b2 ? cur_s0 : 0u);
                    if (64 + lane >= cur_b2 && 64 + lane < B)
                        r1 = X.L.ns[64 + lane] - (64 + lane == cur_b2 ? cur_s0 : 0u);
                    const uint64_t i0 = wave_incl_scan_u64(r0, lane);
                    const uint64_t t0 = __shfl(i0, 63, 64);
                    const uint64_t i1 = wave_incl_scan_u64(r1, lane) + t0;
                    if (lane < B) X.L.P[lane] = i0;
                    if (64 + lane < B) X.L.P[64 + lane] = i1;
                    T = __shfl(i1, 63, 64);
                    wave_sync();
                }
                if (T == 0) break;

                bool had_event = false;
                for (uint64_t q0 = 0; q0 < T && !had_event && !stop; q0 += 64) {
                    const uint64_t q = q0 + (uint64_t)lane;
                    const bool act = q < T;
                    int b2 = cur_b2;
                    uint32_t s = 0;
                    int i1 = 0, i2 = 0;
                    uint32_t p1 = 0, p2 = 0;
                    double v1 = 0, v2 = 0, m1 = 0, m2 = 0;
                    bool acc = false, over = false;
                    if (act) {
                        int lo = cur_b2, hi = B - 1; // smallest b2 with P[b2] > q
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (X.L.P[mid] > q) hi = mid; else lo = mid + 1;
                        }
                        b2 = lo;
                        const uint32_t first = (b2 == cur_b2) ? cur_s0 : 0u;
                        const uint64_t rem = (uint64_t)X.L.ns[b2] - first;
                        s = (uint32_t)(q - (X.L.P[b2] - rem)) + first;
                        const int gB = c2 * B + b2;
                        const bool same = (gA == gB);
                        const int nA = X.L.cnt[gA], nB = X.L.cnt[gB];
                        uint32_t w[4];
                        rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB,
                                  SITE_CAND, s, w);
                        i1 = (int)index_of(w[0], (uint32_t)nA);
                        if (same) {
                            i2 = (int)index_of(w[1], (uint32_t)(nA - 1));
                            if (i2 >= i1) i2++;
                        } else {
                            i2 = (int)index_of(w[1], (uint32_t)nB);
                        }
                        p1 = list_get(X, gA, i1);
                        p2 = list_get(X, gB, i2);
                        v1 = ld_f64(&X.vtot[p1]); v2 = ld_f64(&X.vtot[p2]);
                        m1 = ld_f64(&X.mass[p1]); m2 = ld_f64(&X.mass[p2]);
                        PartProps qa{0, 0, 0, 0, 0}, qb{0, 0, 0, 0, 0};
                        if (use_props) {
                            qa = particle_props(v1, m1, env);
                            qb = particle_props(v2, m2, env);
                        }
                        const double k = kernel_pair(ktype, v1, v2, qa, qb, kp);
                        const double k_max = kmax[b1 * B + b2] * fmax;
                        const double accept_factor = 1.0 / k_max;
                        const double p = (k * factor) * accept_factor;
                        over = p > 1.0;
                        acc = u53(w[2], w[3]) < p;
                    }
                    const uint64_t m = __ballot(act && acc);
                    const int f = m ? __builtin_ctzll(m) : 64;
                    // status of candidates actually reached in sequential order
                    if (__ballot(act && over && lane <= f)) status |= ST_KMAX_EXCEEDED;
                    if (!m) continue;

                    // ---- coagulate the pair held by lane f (uniform from here) ----
                    const int b2f = __shfl(b2, f, 64);
                    const uint32_t sf = __shfl(s, f, 64);
                    int i1f = __shfl(i1, f, 64), i2f = __shfl(i2, f, 64);
                    const uint32_t p1f = __shfl(p1, f, 64), p2f = __shfl(p2, f, 64);
                    const double v1f = __shfl(v1, f, 64), v2f = __shfl(v2, f, 64);
                    const double m1f = __shfl(m1, f, 64), m2f = __shfl(m2, f, 64);
                    const int gB = c2 * B + b2f;
                    const bool same = (gA == gB);

                    // slots up to and including this one have started
                    tot_samp += sum_ns(counted, b2f + 1);
                    counted = b2f + 1;

                    if (n_ev >= st.max_events) { status |= ST_EVENT_OVERFLOW; stop = true; break; }
                    uint32_t w[4];
                    rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB,
                              SITE_COAG, sf, w);
                    const bool rem1 = u53(w[0], w[1]) < nmin / n1;
                    const bool rem2 = u53(w[2], w[3]) < nmin / n2;
                    const double vn = v1f + v2f, mn = m1f + m2f;
                    const bool big1 = (v1f >= v2f);
                    const bool pn_is_1 = (rem1 && rem2) ? big1 : rem1;
                    const uint32_t pn = pn_is_1 ? p1f : p2f, po = pn_is_1 ? p2f : p1f;
                    const bool rem_o = pn_is_1 ? rem2 : rem1;
                    int in = pn_is_1 ? i1f : i2f;
                    const int io = pn_is_1 ? i2f : i1f;
                    const int g_old = pn_is_1 ? gA : gB, g_o = pn_is_1 ? gB : gA;
                    if (vn < st.v_edge[0] || vn >= st.v_edge[B]) status |= ST_OUT_OF_GRID;
                    const int g_new = cc * B + bin_of(st.v_edge, B, vn);
                    if (g_new != g_old && n_app >= st.max_appends) {
                        status |= ST_APPEND_OVERFLOW; stop = true; break;
                    }
                    if (lane == 0)
                        evlog[n_ev] = make_int4((int)pn, (int)po, rem_o ? 1 : 0, cc);
                    n_ev++;
                    if (rem_o) {
                        const int last = list_remove(X, g_o, io, B, lane);
                        if (same && in == last) in = io;
                        wave_sync();
                    }
                    if (g_new != g_old) {
                        list_remove(X, g_old, in, B, lane);
                        wave_sync();
                        list_append(X, g_new, pn, n_app, B);
                        n_app++;
                    }
                    if (lane == 0) { st_f64(&X.vtot[pn], vn); st_f64(&X.mass[pn], mn); }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    wave_sync();
                    tot_coag++;

                    // continue the row after this sample
                    cur_b2 = b2f;
                    cur_s0 = sf + 1;
                    {
                        const int nA = X.L.cnt[gA], nB = X.L.cnt[gB];
                        if (nA < 1 || nB < 1 || (same && nA < 2)) cur_s0 = X.L.ns[cur_b2];
                    }
                    compute_ns(cur_b2 + 1);
                    had_event = true;
                }
                if (!had_event) break;
            }
            if (!stop) tot_samp += sum_ns(counted, B);
        }
    }

    if (lane == 0) {
        st.n_events[c] = n_ev;
        st.n_samp[c] = tot_samp > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)tot_samp;
        st.n_coag[c] = tot_coag;
        if (status) atomicOr(&st.status[c], status);
    }
}

// --------------------------------------------------------------- coag apply
//
// Deferred part of `coagulate` [upstream]: merged particle = species-wise sum
// (aero_particle_coagulate), create times min/max, source histogram summed,
// class set; then caches refreshed from the species and holes closed by
// tail fill (batched form of aero_particle_array's swap-with-last removal).

__device__ __forceinline__ void move_particle(const DeviceState &st, int64_t cb, int c, int dst,
                                              int src)
{
    for (int s = 0; s < st.n_spec; s++)
        st.vol[cb * st.n_spec + (int64_t)s * st.cap + dst] =
            st.vol[cb * st.n_spec + (int64_t)s * st.cap + src];
    for (int s = 0; s < st.n_source; s++)
        st.n_orig[cb * st.n_source + (int64_t)s * st.cap + dst] =
            st.n_orig[cb * st.n_source + (int64_t)s * st.cap + src];
    st.vtot[cb + dst] = st.vtot[cb + src];
    st.mass[cb + dst] = st.mass[cb + src];
    st.wclass[cb + dst] = st.wclass[cb + src];
    st.id[cb + dst] = st.id[cb + src];
    st.lct[cb + dst] = st.lct[cb + src];
    st.gct[cb + dst] = st.gct[cb + src];
}

// Close holes: dead[] bitmap (LDS, one bit per slot) -> the k-th dead slot
// below n_new receives the k-th live slot at or above n_new.  `scratch` is a
// global u32 area of `cap` entries: holes in [0, cap/2), fillers in
// [cap/2, cap) (at most min(n_dead, n_new) <= cap/2 moves).
__device__ void compact_tail_fill(const DeviceState &st, int c, uint64_t *dead, int n_old,
                                  int n_dead, uint32_t *scratch, int lane)
{
    const int64_t cb = (int64_t)c * st.cap;
    const int n_new = n_old - n_dead;
    const int fill_off = st.cap >> 1;
    if (n_dead > 0) {
        const int n_words = (n_old + 63) >> 6;
        uint32_t hole_base = 0, fill_base = 0;
        for (int w0 = 0; w0 < n_words; w0 += 64) {
            const int w = w0 + lane;
            uint64_t hw = 0, fw = 0;
            if (w < n_words) {
                const uint64_t d = dead[w];
                // bits of this word that lie below / at-or-above n_new, below n_old
                const int lo = w << 6;
                uint64_t valid = ~0ull;
                if (n_old - lo < 64) valid = (n_old - lo <= 0) ? 0ull : ((1ull << (n_old - lo)) - 1ull);
                uint64_t below = 0ull;
                if (n_new - lo >= 64) below = ~0ull;
                else if (n_new - lo > 0) below = (1ull << (n_new - lo)) - 1ull;
                hw = d & below & valid;
                fw = ~d & ~below & valid;
            }
            const uint32_t hc = (uint32_t)__popcll(hw), fc = (uint32_t)__popcll(fw);
            const uint32_t hi = wave_incl_scan_u32(hc, lane), fi = wave_incl_scan_u32(fc, lane);
            uint32_t hp = hole_base + hi - hc, fp = fill_base + fi - fc;
            while (hw) { const int b = __builtin_ctzll(hw); hw &= hw - 1; scratch[hp++] = (uint32_t)((w << 6) + b); }
            while (fw) { const int b = __builtin_ctzll(fw); fw &= fw - 1; scratch[fill_off + fp++] = (uint32_t)((w << 6) + b); }
            hole_base += __shfl(hi, 63, 64);
            fill_base += __shfl(fi, 63, 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        const int n_move = (int)hole_base; // == fill_base
        for (int k = lane; k < n_move; k += 64)
            move_particle(st, cb, c, (int)ld_u32(&scratch[k]), (int)ld_u32(&scratch[fill_off + k]));
    }
    if (lane == 0) st.n_part[c] = n_new;
}

__global__ __launch_bounds__(256) void k_coag_apply(DeviceState st)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int c = blockIdx.x * wpb + wave;
    if (c >= st.n_cells) return;
    const int64_t cb = (int64_t)c * st.cap;
    const int n_old = st.n_part[c];
    const int n_ev = st.n_events[c];
    const int n_words_cap = (st.cap + 63) >> 6;
    uint64_t *dead = (uint64_t *)smem + (size_t)wave * n_words_cap;
    const int4 *evlog = st.events + (int64_t)c * st.max_events;
    if (n_ev == 0) return;
    for (int w = lane; w < ((n_old + 63) >> 6); w += 64) dead[w] = 0;
    wave_sync();

    // merge along the log, in order; lanes over species / sources / scalars
    int n_dead = 0;
    const int S = st.n_spec, NS = st.n_source;
    for (int e = 0; e < n_ev; e++) {
        const int4 ev = evlog[e];
        const int pn = ev.x, po = ev.y;
        for (int s = lane; s < S; s += 64) {
            double *a = &st.vol[cb * S + (int64_t)s * st.cap + pn];
            const double b = st.vol[cb * S + (int64_t)s * st.cap + po];
            *a = *a + b;
        }
        for (int s = lane; s < NS; s += 64) {
            int32_t *a = &st.n_orig[cb * NS + (int64_t)s * st.cap + pn];
            *a = *a + st.n_orig[cb * NS + (int64_t)s * st.cap + po];
        }
        if (lane == 0) {
            const double lo = st.lct[cb + po], go = st.gct[cb + po];
            if (lo < st.lct[cb + pn]) st.lct[cb + pn] = lo;
            if (go > st.gct[cb + pn]) st.gct[cb + pn] = go;
            st.wclass[cb + pn] = ev.w;
        }
        if (ev.z) {
            dead[po >> 6] |= (1ull << (po & 63)); // uniform: same value from every lane
            n_dead++;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
    }
    // caches of surviving merged particles from the species
    for (int e = lane; e < n_ev; e += 64) {
        const int pn = evlog[e].x;
        if ((dead[pn >> 6] >> (pn & 63)) & 1ull) continue;
        double vt = 0.0, ms = 0.0;
        for (int s = 0; s < S; s++) {
            const double v = ld_f64(&st.vol[cb * S + (int64_t)s * st.cap + pn]);
            vt = vt + v;
            ms = ms + st.density[s] * v;
        }
        st.vtot[cb + pn] = vt;
        st.mass[cb + pn] = ms;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    compact_tail_fill(st, c, dead, n_old, n_dead, st.perm + cb, lane);
}

// ---------------------------------------------------------------- rebalance
//
// aero_state_rebalance (interface/wrf_pmc_driver.F90:248-250): per class,
// double (copy every particle, new ids, halve the weight) while the class
// holds fewer than n_ideal/2 particles, halve (drop each with probability
// 1/2, double the weight) while it holds more than 2 n_ideal.

__global__ __launch_bounds__(64) void k_rebalance(DeviceState st, int allow_doubling,
                                                  int allow_halving, uint64_t seed, uint32_t step)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int c = blockIdx.x;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = st.cell_id_base + (uint32_t)c;
    uint64_t *dead = (uint64_t *)smem;
    double *mag = st.mag + (int64_t)c * st.n_class;
    const double *n_ideal = st.n_ideal + (int64_t)c * st.n_class;
    int status = 0;
    for (int cl = 0; cl < st.n_class; cl++) {
        for (uint32_t iter = 0; iter < 32; iter++) {
            const int n = st.n_part[c];
            uint32_t cnt = 0;
            for (int i = lane; i < n; i += 64) cnt += (st.wclass[cb + i] == cl);
            const int ncl = (int)wave_sum_u64(cnt);
            const double ideal = n_ideal[cl];
            if (allow_doubling && ncl > 0 && (double)ncl < ideal / 2.0) {
                if (n + ncl > st.cap) { status |= ST_CAP_OVERFLOW; break; }
                const int64_t nid = st.next_id[c];
                uint32_t base = 0;
                for (int i0 = 0; i0 < n; i0 += 64) {
                    const int i = i0 + lane;
                    const bool is = (i < n) && (st.wclass[cb + i] == cl);
                    const uint64_t m = __ballot(is);
                    const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (is) {
                        move_particle(st, cb, c, n + (int)k, i);
                        st.id[cb + n + k] = nid + (int64_t)k;
                    }
                    base += (uint32_t)__popcll(m);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                if (lane == 0) {
                    st.next_id[c] = nid + ncl;
                    st.n_part[c] = n + ncl;
                    mag[cl] = mag[cl] / 2.0;
                }
                __threadfence_block();
                wave_sync();
            } else if (allow_halving && (double)ncl > ideal * 2.0) {
                uint32_t nd = 0;
                for (int w0 = 0; w0 < ((n + 63) >> 6); w0++) {
                    const int i = (w0 << 6) + lane;
                    bool drop = false;
                    if (i < n && st.wclass[cb + i] == cl) {
                        uint32_t w[4];
                        rng_block(seed, cell_id, step, (uint32_t)i, SITE_HALVE,
                                  (uint32_t)cl * 32u + iter, w);
                        drop = u53(w[0], w[1]) < 0.5;
                    }
                    const uint64_t m = __ballot(drop);
                    dead[w0] = m; // uniform store
                    nd += (uint32_t)__popcll(m);
                }
                wave_sync();
                compact_tail_fill(st, c, dead, n, (int)nd, st.perm + cb, lane);
                if (lane == 0) mag[cl] = mag[cl] * 2.0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __threadfence_block();
                wave_sync();
            } else {
                break;
            }
        }
    }
    if (lane == 0 && status) atomicOr(&st.status[c], status);
}

// -------------------------------------------------------------- diagnostics
//
// partmc_process observables (interface/wrf_pmc_driver.F90:783,838-857):
// per cell tot_num_conc, tot_mass_conc, and the n_dbin histograms of
// num_conc and num_conc*mass over dry diameter divided by the bin log-width.
// Number quantities are accumulated as integer counts per (class, bin) and
// converted once, so they are exact; mass quantities are fp64 tree sums.

__global__ __launch_bounds__(256) void k_diagnostics(DeviceState st, double *tot_num,
                                                     double *tot_mass, double *num_dist,
                                                     double *mass_dist)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int c = blockIdx.x, t = threadIdx.x;
    const int nd = st.n_dbin, C = st.n_class, S = st.n_spec;
    double *edge = (double *)smem;                       // [nd+1]
    double *msum = edge + (nd + 1);                      // [nd]
    double *red = msum + nd;                             // [256]
    uint32_t *ccnt = (uint32_t *)(red + 256);            // [C][nd]
    uint32_t *clcnt = ccnt + (size_t)C * nd;             // [C]
    for (int j = t; j <= nd; j += 256) edge[j] = st.d_edge[j];
    for (int j = t; j < nd; j += 256) msum[j] = 0.0;
    for (int j = t; j < C * nd; j += 256) ccnt[j] = 0;
    for (int j = t; j < C; j += 256) clcnt[j] = 0;
    __syncthreads();
    const int n = st.n_part[c];
    const int64_t cb = (int64_t)c * st.cap;
    const double *mag = st.mag + (int64_t)c * C;
    double tm = 0.0;
    for (int i = t; i < n; i += 256) {
        double vdry = 0.0, ms = 0.0;
        for (int s = 0; s < S; s++) {
            const double v = st.vol[cb * S + (int64_t)s * st.cap + i];
            if (s != st.i_water) vdry = vdry + v;
            ms = ms + st.density[s] * v;
        }
        const int cl = st.wclass[cb + i];
        const double nc = mag[cl];
        const double diam = 2.0 * det_cbrt(vdry * (0.75 / PI));
        atomicAdd(&clcnt[cl], 1u);
        tm += ms * nc;
        if (diam >= edge[0] && diam < edge[nd]) {
            const int b = bin_of(edge, nd, diam);
            atomicAdd(&ccnt[cl * nd + b], 1u);
            atomicAdd(&msum[b], nc * ms);
        }
    }
    red[t] = tm;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) red[t] += red[t + d];
        __syncthreads();
    }
    if (t == 0) {
        double tn = 0.0;
        for (int cl = 0; cl < C; cl++) tn = tn + (double)clcnt[cl] * mag[cl];
        tot_num[c] = tn;
        tot_mass[c] = red[0];
    }
    for (int b = t; b < nd; b += 256) {
        double x = 0.0;
        for (int cl = 0; cl < C; cl++) x = x + (double)ccnt[cl * nd + b] * mag[cl];
        num_dist[(int64_t)c * nd + b] = x / st.d_width[b];
        mass_dist[(int64_t)c * nd + b] = msum[b] / st.d_width[b];
    }
}

// ---------------------------------------------------------------- generator
//
// Synthetic log-normal mode mixture (shape of aero_state_add_aero_dist_sample,
// interface/wrf_pmc_init.F90:1063,1764-1777), one particle per thread.

__global__ void k_generate(DeviceState st, GenSpecDev g, int n, uint64_t seed, double create_time)
{
    const int c = blockIdx.y;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = st.cell_id_base + (uint32_t)c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t w[4];
        rng_block(seed, cell_id, 0u, (uint32_t)i, SITE_GEN, 0, w);
        const double um = (double)w[0] * 0x1.0p-32;
        int mode = 0;
        while (mode < g.n_mode - 1 && um >= g.mode_frac[mode]) mode++;
        const double z = det_norminv(u53(w[2], w[3]));
        const double r = det_exp(g.ln_gm_radius[mode] + g.ln_gsd[mode] * z);
        const double v = (4.0 / 3.0) * PI * r * r * r;
        double vt = 0.0, ms = 0.0;
        for (int s = 0; s < st.n_spec; s++) {
            const double vs = g.vol_frac[mode * st.n_spec + s] * v;
            st.vol[cb * st.n_spec + (int64_t)s * st.cap + i] = vs;
            vt = vt + vs;
            ms = ms + st.density[s] * vs;
        }
        for (int s = 0; s < st.n_source; s++)
            st.n_orig[cb * st.n_source + (int64_t)s * st.cap + i] = (s == g.mode_source[mode]) ? 1 : 0;
        st.vtot[cb + i] = vt;
        st.mass[cb + i] = ms;
        st.wclass[cb + i] = g.mode_class[mode];
        st.id[cb + i] = ((int64_t)cell_id << 32) + i;
        st.lct[cb + i] = create_time;
        st.gct[cb + i] = create_time;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st.n_part[c] = n;
}

// ------------------------------------------------------------------- totals

__global__ __launch_bounds__(256) void k_totals(DeviceState st, int64_t *out /*[4]*/)
{
    __shared__ int64_t r0[256], r1[256], r2[256];
    __shared__ int r3[256];
    const int t = threadIdx.x;
    int64_t a = 0, b = 0, p = 0;
    int s = 0;
    for (int c = blockIdx.x * 256 + t; c < st.n_cells; c += gridDim.x * 256) {
        a += st.n_samp[c]; b += st.n_coag[c]; p += st.n_part[c]; s |= st.status[c];
    }
    r0[t] = a; r1[t] = b; r2[t] = p; r3[t] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) { r0[t] += r0[t + d]; r1[t] += r1[t + d]; r2[t] += r2[t + d]; r3[t] |= r3[t + d]; }
        __syncthreads();
    }
    if (t == 0) {
        atomicAdd((unsigned long long *)&out[0], (unsigned long long)r0[0]);
        atomicAdd((unsigned long long *)&out[1], (unsigned long long)r1[0]);
        atomicAdd((unsigned long long *)&out[2], (unsigned long long)r2[0]);
        atomicOr((int *)&out[3], r3[0]);
    }
}

// --------------------------------------------------------------- test hooks

__global__ void k_test_math(int op, int64_t n, const double *a, const double *b, const double *c3,
                            const double *c4, int ktype, double temp, double pressure,
                            double const_k, double additive_beta, double *out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = 0.0;
    switch (op) {
    case 0: r = det_exp(a[i]); break;
    case 1: r = det_log(a[i]); break;
    case 2: r = det_cbrt(a[i]); break;
    case 3: r = det_norminv(a[i]); break;
    case 4: r = __builtin_sqrt(a[i]); break;
    case 5: r = a[i] / b[i]; break;
    case 6: r = (double)poisson(a[i], b[i]); break;
    case 7: {
        const EnvDerived env = env_derive(temp, pressure);
        const KernelParams kp{const_k, additive_beta};
        PartProps qa{0, 0, 0, 0, 0}, qb{0, 0, 0, 0, 0};
        if (kernel_needs_props(ktype)) {
            qa = particle_props(a[i], b[i], env);
            qb = particle_props(c3[i], c4[i], env);
        }
        r = kernel_pair(ktype, a[i], c3[i], qa, qb, kp);
        break;
    }
    default: break;
    }
    out[i] = r;
}

__global__ void k_test_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                              uint32_t k1, uint32_t *out)
{
    uint32_t w[4] = {c0, c1, c2, c3};
    philox4x32_10(w, k0, k1);
    out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = w[3];
}

// ------------------------------------------------------------------ launchers

size_t bin_sort_lds_bytes(const DeviceState &st)
{
    const int B = st.n_bin, C = st.n_class, G = B * C, D = B > C ? B : C;
    size_t o = sizeof(double) * (size_t)(B + 1) + sizeof(uint32_t) * (size_t)D;
    if (C > 1) o += sizeof(uint32_t) * (size_t)G;
    o += sizeof(uint16_t) * 256 * (size_t)D;
    if (C > 1) o += sizeof(uint16_t) * (size_t)st.cap;
    o += (size_t)st.cap;
    if (C > 1) o += (size_t)st.cap;
    return (o + 15) & ~(size_t)15;
}
size_t sampler_lds_bytes_per_cell(const DeviceState &st)
{
    return sampler_lds_bytes(st.n_bin * st.n_class, st.n_bin, st.n_class, st.max_appends);
}
size_t diagnostics_lds_bytes(const DeviceState &st)
{
    return sizeof(double) * (size_t)(2 * st.n_dbin + 1 + 256) +
           sizeof(uint32_t) * (size_t)(st.n_class * st.n_dbin + st.n_class);
}

// dynamic LDS above 64 KiB needs the per-function opt-in
template <typename K>
static inline void allow_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

static inline int waves_per_block(size_t lds_per_cell)
{
    int wpb = 4;
    while (wpb > 1 && lds_per_cell * wpb > 64 * 1024) wpb >>= 1;
    return wpb;
}

static inline dim3 grid2(int64_t n_max, int n_y)
{
    int64_t gx = (n_max + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    return dim3((unsigned)gx, (unsigned)n_y);
}

hipError_t launch_scatter_f64(const double *src, double *dst, const int64_t *part_offset, int n_up,
                              int cap, int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_scatter_f64, grid2(cap, n_up), dim3(256), 0, s, src, dst, part_offset,
                       n_up, cap, dst_cell_stride, src_row_stride, n_rows, dst_row_stride);
    return hipGetLastError();
}
hipError_t launch_scatter_i32(const int32_t *src, int32_t *dst, const int64_t *part_offset,
                              int n_up, int cap, int64_t dst_cell_stride, int64_t src_row_stride,
                              int n_rows, int64_t dst_row_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_scatter_i32, grid2(cap, n_up), dim3(256), 0, s, src, dst, part_offset,
                       n_up, cap, dst_cell_stride, src_row_stride, n_rows, dst_row_stride);
    return hipGetLastError();
}
hipError_t launch_gather_f64(double *dst, const double *src, const int64_t *part_offset, int n_dn,
                             int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_f64, grid2(src_cell_stride, n_dn), dim3(256), 0, s, dst, src,
                       part_offset, src_cell_stride, dst_row_stride, n_rows, src_row_stride);
    return hipGetLastError();
}
hipError_t launch_gather_i32(int32_t *dst, const int32_t *src, const int64_t *part_offset,
                             int n_dn, int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_i32, grid2(src_cell_stride, n_dn), dim3(256), 0, s, dst, src,
                       part_offset, src_cell_stride, dst_row_stride, n_rows, src_row_stride);
    return hipGetLastError();
}
hipError_t launch_set_n_part(int32_t *n_part, const int64_t *part_offset, int n_up, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_n_part, dim3((n_up + 255) / 256), dim3(256), 0, s, n_part,
                       part_offset, n_up);
    return hipGetLastError();
}
hipError_t launch_refresh_caches(const DeviceState &st, int cell_first, int n_cells_do,
                                 hipStream_t s)
{
    hipLaunchKernelGGL(k_refresh_caches, grid2(st.cap, n_cells_do), dim3(256), 0, s, st,
                       cell_first, n_cells_do);
    return hipGetLastError();
}
hipError_t launch_kernel_bounds(const DeviceState &st, int ktype, hipStream_t s)
{
    const size_t lds = sizeof(PartProps) * (size_t)st.n_bin * 6;
    hipLaunchKernelGGL(k_kernel_bounds, dim3(st.n_cells), dim3(256), lds, s, st, ktype);
    return hipGetLastError();
}
hipError_t launch_bin_sort(const DeviceState &st, hipStream_t s)
{
    allow_lds(k_bin_sort, bin_sort_lds_bytes(st));
    hipLaunchKernelGGL(k_bin_sort, dim3(st.n_cells), dim3(256), bin_sort_lds_bytes(st), s, st);
    return hipGetLastError();
}
hipError_t launch_coag_sample(const DeviceState &st, int ktype, double del_t, uint64_t seed,
                              uint32_t step, hipStream_t s)
{
    const size_t per = sampler_lds_bytes_per_cell(st);
    const int wpb = waves_per_block(per);
    allow_lds(k_coag_sample, per * wpb);
    hipLaunchKernelGGL(k_coag_sample, dim3((st.n_cells + wpb - 1) / wpb), dim3(64 * wpb),
                       per * wpb, s, st, ktype, del_t, seed, step, per);
    return hipGetLastError();
}
hipError_t launch_coag_apply(const DeviceState &st, hipStream_t s)
{
    const size_t per = sizeof(uint64_t) * (size_t)((st.cap + 63) >> 6);
    const int wpb = waves_per_block(per);
    hipLaunchKernelGGL(k_coag_apply, dim3((st.n_cells + wpb - 1) / wpb), dim3(64 * wpb), per * wpb,
                       s, st);
    return hipGetLastError();
}
hipError_t launch_rebalance(const DeviceState &st, int allow_doubling, int allow_halving,
                            uint64_t seed, uint32_t step, hipStream_t s)
{
    const size_t lds = sizeof(uint64_t) * (size_t)((st.cap + 63) >> 6);
    hipLaunchKernelGGL(k_rebalance, dim3(st.n_cells), dim3(64), lds, s, st, allow_doubling,
                       allow_halving, seed, step);
    return hipGetLastError();
}
hipError_t launch_diagnostics(const DeviceState &st, double *tot_num, double *tot_mass,
                              double *num_dist, double *mass_dist, hipStream_t s)
{
    hipLaunchKernelGGL(k_diagnostics, dim3(st.n_cells), dim3(256), diagnostics_lds_bytes(st), s, st,
                       tot_num, tot_mass, num_dist, mass_dist);
    return hipGetLastError();
}
hipError_t launch_generate(const DeviceState &st, const GenSpecDev &g, int n, uint64_t seed,
                           double create_time, hipStream_t s)
{
    hipLaunchKernelGGL(k_generate, grid2(n, st.n_cells), dim3(256), 0, s, st, g, n, seed,
                       create_time);
    return hipGetLastError();
}
hipError_t launch_totals(const DeviceState &st, int64_t *out4, hipStream_t s)
{
    int g = (st.n_cells + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_totals, dim3(g), dim3(256), 0, s, st, out4);
    return hipGetLastError();
}
hipError_t launch_test_math(int op, int64_t n, const double *a, const double *b, const double *c3,
                            const double *c4, int ktype, double temp, double pressure,
                            double const_k, double additive_beta, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op, n, a, b,
                       c3, c4, ktype, temp, pressure, const_k, additive_beta, out);
    return hipGetLastError();
}
hipError_t launch_test_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t *out_dev,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_test_philox, dim3(1), dim3(1), 0, s, ctr[0], ctr[1], ctr[2], ctr[3],
                       key[0], key[1], out_dev);
    return hipGetLastError();
}

} // namespace pmcx
