// pmcx_kernels.hip -- gfx950 kernels of the PartMC per-cell microphysics path.
//
// Path replaced (one Fortran loop body per grid cell, serial per MPI rank):
//   partmc_timestep                 interface/wrf_pmc_driver.F90:169-254
//     mc_coag(COAG_KERNEL_TYPE_BROWN, ...)               driver:192-194
//     aero_state_rebalance                               driver:248-250
//   partmc_process observables                           driver:783,838-857
//
// Layout in HBM (structure of arrays, one slab of `cap` slots per cell):
//   vol   [n_cells][cap][n_spec] f64   particle%vol(:): a particle's species are contiguous,
//                                      a coagulation touches 3 lines per particle at 20 species
//   hot   [n_cells][cap] 32 B          {cached sum_s vol, cached sum_s density_s*vol, least,
//                                      greatest create time}: what a pair evaluation AND a
//                                      coagulation read and write of a particle: one line
//   cold  [n_cells][cap] 16 B          {id, source-0 count}: whole-particle moves only (and
//                                      events when sources are tracked)
//   wclass[n_cells][cap] i32
//   n_orig[n_cells][n_source-1][cap] i32  constituents from sources 1..
// Kernels:
//   k_bin_sort     one 256-thread workgroup per cell; stable counting sort of
//                  the slots by (class, bin) -> perm, gcnt, goff.  Streams
//                  vtot once: the HBM-bound pass (12 B per particle).
//   k_coag_lane    the pair sampler for large blocks of cells: ONE LANE PER CELL, every
//                  lane runs the sequential PartMC sweep of its own cell; at the full domain
//                  bound by the random 64-byte lines its lists ask for.
//   k_coag_sample  the pair sampler for small blocks (or many classes): 8, 16 or 64 lanes
//                  per cell; candidate pairs of the sweep evaluated a window at a time, first
//                  accepted pair found by ballot, applied, later slots redrawn.
//   k_compact      one wavefront per cell; stable compaction of cells whose
//                  tombstone fraction exceeds a threshold.
//   k_rebalance, k_diagnostics, k_kernel_bounds, k_generate: full-population
//                  or per-cell table passes around the above.
// All arithmetic that feeds a decision is fp64 with contraction off.
#include "pmcx_kernels.h"
#include "pmcx_device_math.h"

#include <cstdlib>

#ifndef PMCX_WIN
#define PMCX_WIN 8 // partner slots per sampler window (lanes-per-cell < 64)
#endif

namespace pmcx {

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Same for LDS traffic only: a wavefront's LDS operations are served in order, so its lanes
// see each other's LDS writes once the counter of outstanding LDS operations is zero; global
// stores in flight are not waited for (wave_sync's fence waits for their acknowledgement).
__device__ __forceinline__ void lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
__device__ __forceinline__ uint32_t ld_u16(const uint16_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u16(uint16_t *p, uint32_t v)
{
    __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the global id of local cell c: the key of every random draw (results do not depend on how the
// domain is cut into blocks)
__device__ __forceinline__ uint32_t cell_gid(const DeviceState &st, int c)
{
    if (st.id_nx_local == 0) return st.cell_id_base + (uint32_t)c;
    return st.cell_id_base + (uint32_t)(c / st.id_nx_local) * (uint32_t)st.id_nx_global + (uint32_t)(c % st.id_nx_local);
}

// particles of source s among the constituents of slot i: the first source sits in the
// particle's metadata record (it is read and written with the create times), the others in
// their own planes
__device__ __forceinline__ int32_t &norig(const DeviceState &st, int64_t cb, int s, int64_t i)
{
    if (s == 0) return st.cold[cb + i].n_orig0;
    return st.n_orig[cb * (st.n_source - 1) + (int64_t)(s - 1) * st.cap + i];
}

// particle%component(:), see DeviceState::comp_src (at, dst, src: cell base + slot)
__device__ __forceinline__ void comp_copy(const DeviceState &st, int64_t dst, int64_t src)
{
    if (!st.n_comp_max) return;
    const int K = st.n_comp_max;
    st.comp_src[dst] = st.comp_src[src];
    for (int k = 0; k < K; k++) st.comp_time[dst * K + k] = st.comp_time[src * K + k];
}
__device__ __forceinline__ void comp_single(const DeviceState &st, int64_t at, int source, double t)
{
    if (!st.n_comp_max) return;
    const int K = st.n_comp_max;
    st.comp_src[at] = ~0xFFull | (uint64_t)(source & 0xFF);
    st.comp_time[at * K] = t;
    for (int k = 1; k < K; k++) st.comp_time[at * K + k] = 0.0;
}
__device__ __forceinline__ int comp_count(uint64_t w)
{
    int n = 0;
    for (int k = 0; k < 8; k++) n += ((w >> (8 * k)) & 0xFFu) != 0xFFu;
    return n;
}

// The cells the loop nest of partmc_timestep covers (interface/wrf_pmc_driver.F90:157-171: every
// owned cell when the domain is doubly periodic, otherwise the interior 2 .. n-1 in x and y):
// pmcx_set_loop_cells.  A cell outside it is left alone by mc_coag, the scenario updates and the
// rebalance (its lists may still be sorted: that changes nothing a result depends on).
__device__ __forceinline__ bool in_loop(const DeviceState &st, int c)
{
    return !st.loop_mask || st.loop_mask[c] != 0;
}

// tombstone: the slot stays allocated, every consumer skips it (live_class: a negative cached
// total volume is the one mark; wclass of a dead slot is unspecified)
__device__ __forceinline__ void mark_dead(const DeviceState &st, int64_t cb, int i)
{
    st_f64(&st.hot[cb + i].vtot, -1.0);
    st.gkey[cb + i] = (uint16_t)0xFFFFu; // (the slot's sort key: DeviceState::gkey)
}

// Class of the particle in slot i, or -1 for a tombstone.  A slot is dead when its cached
// total volume is negative (the lane-per-cell sampler marks a removed particle there, in the
// record it has just read, and leaves wclass alone: a store to an untouched line per event
// costs it 5 % of its time); wclass of a dead slot is unspecified.
__device__ __forceinline__ int live_class(const DeviceState &st, int64_t cb, int i)
{
    if (st.hot[cb + i].vtot < 0.0) return -1;
    return st.n_class > 1 ? st.wclass[cb + i] : 0;
}
// same, for kernels that kill particles themselves before looking again
__device__ __forceinline__ int live_class_c(const DeviceState &st, int64_t cb, int i)
{
    if (ld_f64(&st.hot[cb + i].vtot) < 0.0) return -1;
    return st.n_class > 1 ? (int32_t)ld_u32((const uint32_t *)&st.wclass[cb + i]) : 0;
}
// volume of species s of slot i.  With ONE species the plane `vol` is not kept current by the
// lane-per-cell sampler: that volume and the cached total are sums of the same terms in the
// same order, so the cached total is read instead (and `vol` is written at upload, generation
// and emission only).
__device__ __forceinline__ double spec_vol(const DeviceState &st, int64_t cb, int s, int i)
{
    if (st.n_spec == 1) return st.hot[cb + i].vtot;
    return st.vol[(cb + i) * st.n_spec + s];
}

// Species s of the particle whose record starts at rec (a particle's species are
// contiguous).  Full-population passes walk a record with 16-byte loads when the species
// count is even: species_pair(rec, s) returns species s and s + 1 for even s.
__device__ __forceinline__ double2 species_pair(const double *rec, int s)
{
    return *(const double2 *)(rec + s);
}

// Species of 64 consecutive particles staged through LDS, for the memory-bound passes over
// the whole population: the records of consecutive particles are contiguous, so the
// wavefront loads them coalesced (element j of the block to tile[p * (S + 1) + s]) and each
// lane then reads its own record tile[lane * (S + 1) + s] (the odd row length keeps the
// lanes on different banks).  Every lane of the wavefront must call it; i0 is a multiple of 64.
__host__ __device__ inline size_t species_tile_bytes(int n_spec, int waves)
{
    return n_spec > 1 ? sizeof(double) * 64 * (size_t)(n_spec + 1) * (size_t)waves : 0;
}
__device__ __forceinline__ void stage_species(const DeviceState &st, int64_t cb, int i0, int n,
                                              double *tile, int lane)
{
    const int S = st.n_spec;
    int np = n - i0;
    np = np > 64 ? 64 : (np < 0 ? 0 : np);
    const int cnt = np * S;
    const double *src = st.vol + (cb + i0) * S;
    wave_sync(); // the previous tile has been consumed
    if ((S & 1) == 0) {
        // an even species count (the reference's 20): 16 bytes a lane and request, four requests in flight
        // (a pair of species never spans two particles; one 8-byte request at a time left the 160-byte
        // records streaming at 2.4 TB/s)
        const double2 *src2 = (const double2 *)src;
        const int cnt2 = cnt >> 1;
        int p = (2 * lane) / S, sp = 2 * lane - p * S;
        const int q2 = 128 / S, r2 = 128 - q2 * S;
        for (int j0 = 0; j0 < cnt2; j0 += 256) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = j0 + u * 64 + lane;
                v[u] = jj < cnt2 ? src2[jj] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j0 + u * 64 + lane < cnt2) {
                    tile[p * (S + 1) + sp] = v[u].x;
                    tile[p * (S + 1) + sp + 1] = v[u].y;
                }
                p += q2; sp += r2;
                if (sp >= S) { sp -= S; p++; }
            }
        }
    } else {
        int p = lane / S, sp = lane - p * S;
        const int q = 64 / S, r = 64 - q * S;
        for (int j = lane; j < cnt; j += 64) {
            tile[p * (S + 1) + sp] = src[j];
            p += q; sp += r;
            if (sp >= S) { sp -= S; p++; }
        }
    }
    wave_sync();
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

// Exact bin of a volume on the log-spaced grid.  A float estimate of log2(v)
// from the exponent and the top mantissa bits lands within one bin of the
// answer (host checks the grid is coarse enough: st.bin_fast); one compare
// against each edge of that bin settles it.  Falls back to binary search.
__device__ __forceinline__ int bin_index(const double *v_edge, int n_bin, double v,
                                         const DeviceState &st)
{
    if (!st.bin_fast) return bin_of(v_edge, n_bin, v);
    const uint32_t hi = (uint32_t)(d2u(v) >> 32);
    const float l2 = (float)((int)((hi >> 20) & 0x7FF) - 1023) + (float)(hi & 0xFFFFFu) * 0x1.0p-20f;
    int b = (int)((l2 - st.log2_v0) * st.inv_dlog2);
    b = b < 0 ? 0 : (b > n_bin - 1 ? n_bin - 1 : b);
    const double e_lo = v_edge[b], e_hi = v_edge[b + 1];
    b += (int)(v >= e_hi) - (int)(v < e_lo);
    return b < 0 ? 0 : (b > n_bin - 1 ? n_bin - 1 : b);
}

// Brings the sort keys of cell c up to date: the slots the pair sampler re-binned or removed
// since the last sort are in the cell's event log (DeviceState::evlog), applied here in order (a
// slot may appear more than once: one lane stores, entry by entry).  Every kernel that changes a
// cell's particles and keeps its keys current calls this first; one wavefront, all lanes.
__device__ __forceinline__ void apply_event_log(const DeviceState &st, int c, int lane)
{
    if (st.key_valid[c] == 0) return;
    const int nl = min(st.n_log[c], EVLOG_CAP);
    if (nl <= 0) return;
    const int64_t cb = (int64_t)c * st.cap;
    const int n_hi = st.n_hi[c];
    for (int j0 = 0; j0 < nl; j0 += 64) {
        const uint32_t e = j0 + lane < nl ? st.evlog[(int64_t)c * EVLOG_CAP + j0 + lane] : 0u;
        const int m = min(64, nl - j0);
        for (int l = 0; l < m; l++) {
            const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
            if (lane == 0 && (int)(x & 0xFFFFu) < n_hi) st.gkey[cb + (x & 0xFFFFu)] = (uint16_t)(x >> 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st.n_log[c] = 0;
    wave_sync();
}


// Live particles of every weight class of cell c (several classes), into cls[0 .. n_class) in LDS, by one
// wavefront: the lists' lengths while they hold; else ONE pass over the slots' keys, eight to a 16-byte
// load (class = key / n_bin), or over the records where the keys do not hold -- counted with LDS atomics
// (a handful of addresses a few lanes share: eight copies of each counter, a lane adds to copy lane & 7; cls
// holds 8 n_class ints, the sums end up in its first n_class).  Until round 4 the transport's probabilities, the
// preweight and the rebalance each walked the cell once PER CLASS or issued a ballot per class and chunk
// with one load in flight: 8 + 21 + 10 ms of a coupled step at ten classes and 122 880 cells.
__device__ __forceinline__ void wave_class_counts(const DeviceState &st, int c, int lane, int *cls)
{
    const int C = st.n_class, B = st.n_bin;
    for (int q = lane; q < 8 * C; q += 64) cls[q] = 0;
    lds_sync();
    int *mine = cls + (lane & 7) * C;   // (this lane's copy of the counters)
    if (st.sort_valid[c]) {
        const int32_t *g = st.gcnt + (int64_t)c * C * B;
        for (int x = lane; x < C * B; x += 64) {
            const int n = g[x];
            if (n) atomicAdd(&mine[x / B], n);
        }
    } else {
        const int64_t cb = (int64_t)c * st.cap;
        const int n_hi = st.n_hi[c];
        if (st.key_valid[c] != 0 && (st.cap & 7) == 0) {
            const uint32_t mdiv = (1u << 24) / (uint32_t)B + 1u;  // key / B = key * mdiv >> 24 (keys < 65 535, B <= 255)
            const uint4 *k4 = (const uint4 *)(st.gkey + cb);
            const int n8 = (n_hi + 7) >> 3;
            for (int j0 = 0; j0 < n8; j0 += 128) {
                uint4 v[2];
#pragma unroll
                for (int u = 0; u < 2; u++) v[u] = j0 + u * 64 + lane < n8 ? k4[j0 + u * 64 + lane] : make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    const int i0 = (j0 + u * 64 + lane) * 8;
#pragma unroll
                    for (int h = 0; h < 8; h++) {
                        const uint32_t key = (w4[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                        if (key != 0xFFFFu && i0 + h < n_hi) atomicAdd(&mine[(key * mdiv) >> 24], 1);
                    }
                }
            }
        } else {
            for (int i0 = 0; i0 < n_hi; i0 += 256) {
                int wc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) wc[u] = i0 + u * 64 + lane < n_hi ? live_class_c(st, cb, i0 + u * 64 + lane) : -1;
#pragma unroll
                for (int u = 0; u < 4; u++) if (wc[u] >= 0) atomicAdd(&mine[wc[u]], 1);
            }
        }
    }
    lds_sync();
    for (int q = lane; q < C; q += 64) {
        int t = 0;
        for (int r = 0; r < 8; r++) t += cls[r * C + q];
        cls[q] = t;    // (copy 0's place: read before it is written, by the same lane)
    }
    lds_sync();
}

// the same for every cell, into cls_cnt [n_cells][n_class] (the transport's probabilities and preweight)
__global__ __launch_bounds__(256) void k_class_counts(DeviceState st, int32_t *cls_cnt)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C = st.n_class;
    int *cls = (int *)smem + (size_t)wave * 8 * C;
    for (int c = blockIdx.x * 4 + wave; c < st.n_cells; c += gridDim.x * 4) {
        apply_event_log(st, c, lane);
        wave_class_counts(st, c, lane, cls);
        for (int q = lane; q < C; q += 64) cls_cnt[(int64_t)c * C + q] = cls[q];
        lds_sync();
    }
}

// sort key of a live particle: its group class * n_bin + bin
__device__ __forceinline__ uint16_t key_of(const DeviceState &st, double vtot, int cl)
{
    return (uint16_t)(cl * st.n_bin + bin_index(st.v_edge, st.n_bin, vtot, st));
}

// ------------------------------------------------------------ upload/download

// packed [n_total] (cell-major) -> slab [n_cells][cap]
// (es = stride between consecutive particles in the slab, in elements: 1 for a plane,
// the record length for one field of a record array)
__global__ void k_scatter_f64(const double *__restrict__ src, double *__restrict__ dst,
                              const int64_t *__restrict__ part_offset, int n_up, int cap,
                              int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride, int es)
{
    // destination element of (cell k, row r, particle i): k * cell_stride + r * row_stride + i * es
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[(int64_t)k * dst_cell_stride + r * dst_row_stride + i * es] =
                src[r * src_row_stride + o0 + i];
}
__global__ void k_scatter_i32(const int32_t *__restrict__ src, int32_t *__restrict__ dst,
                              const int64_t *__restrict__ part_offset, int n_up, int cap,
                              int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride, int es)
{
    // destination element of (cell k, row r, particle i): k * cell_stride + r * row_stride + i * es
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[(int64_t)k * dst_cell_stride + r * dst_row_stride + i * es] =
                src[r * src_row_stride + o0 + i];
}
__global__ void k_gather_f64(double *__restrict__ dst, const double *__restrict__ src,
                             const int64_t *__restrict__ part_offset, int64_t src_cell_stride,
                             int64_t dst_row_stride, int n_rows, int64_t src_row_stride, int es)
{
    // source element of (cell k, row r, particle i): k * cell_stride + r * row_stride + i * es
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[r * dst_row_stride + o0 + i] =
                src[(int64_t)k * src_cell_stride + r * src_row_stride + i * es];
}
__global__ void k_gather_i32(int32_t *__restrict__ dst, const int32_t *__restrict__ src,
                             const int64_t *__restrict__ part_offset, int64_t src_cell_stride,
                             int64_t dst_row_stride, int n_rows, int64_t src_row_stride, int es)
{
    // source element of (cell k, row r, particle i): k * cell_stride + r * row_stride + i * es
    const int k = blockIdx.y;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0;
    for (int r = 0; r < n_rows; r++)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            dst[r * dst_row_stride + o0 + i] =
                src[(int64_t)k * src_cell_stride + r * src_row_stride + i * es];
}
__global__ void k_set_n_part(int32_t *n_part, int32_t *n_hi, const int64_t *part_offset, int n_up)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_up) {
        n_part[k] = (int32_t)(part_offset[k + 1] - part_offset[k]);
        n_hi[k] = n_part[k];
    }
}

// vtot/mass from the species volumes (sequential sum over species)
__global__ void k_refresh_caches(DeviceState st, int cell_first, int n_cells_do)
{
    const int c = cell_first + blockIdx.y;
    const int n = st.n_part[c];
    const int64_t cb = (int64_t)c * st.cap;
    const int S = st.n_spec;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double *rec = st.vol + (cb + i) * S;
        double vt = 0.0, ms = 0.0;
        int s = 0;
        if ((S & 1) == 0)
            for (; s < S; s += 2) {
                const double2 v = species_pair(rec, s);
                vt = vt + v.x; ms = ms + st.density[s] * v.x;
                vt = vt + v.y; ms = ms + st.density[s + 1] * v.y;
            }
        for (; s < S; s++) {
            const double v = rec[s];
            vt = vt + v;
            ms = ms + st.density[s] * v;
        }
        st.hot[cb + i].vtot = vt;
        st.hot[cb + i].mass = ms;
    }
}

// ------------------------------------------------------------ kernel bounds
//
// est_k_minmax_binned_unweighted [upstream]: per cell, K_max(b1,b2) = max of K
// over 3x3 sample volumes of the two bins and (rho_min|rho_max)^2.
//
// The upper edge of a bin is the lower edge of the next, so the 3 B sample volumes are
// 2 B + 1 distinct ones (lattice point l = 2 b + s), and K is symmetric bit for bit
// (brown_pair, sedi_pair: commutative sums and products only): K is evaluated once per
// unordered pair of lattice points -- over the density pairs, one when every species has the
// same density -- rounded up to fp32 into a triangular table in LDS, and a bin pair takes the
// maximum of its 3x3 block (rounding up is monotone: the maximum of the rounded values is the
// rounded maximum).  About 1/9 of the evaluations of the direct form with one density, 4/9
// with two.

__host__ __device__ inline size_t kernel_bounds_lds_bytes(int n_bin)
{
    const size_t L = 2 * (size_t)n_bin + 1;
    return sizeof(double) * (5 * L * 2 + L + 1) + sizeof(float) * (L * (L + 1) / 2);
}

// Which cells need new bounds (after a sort: the lists are current).  A cell whose bounds were
// computed for this kernel, at this temperature and pressure, over a range of bins that still
// covers its populated bins plus the two a merged pair can reach, is left alone; any other is
// listed, with the range [first populated bin, last + margin] it will get (everything else in
// its table is stale: the samplers raise ST_KMAX_RANGE should a particle ever be created
// beyond that range within the call).  One lane per cell.
constexpr int BOUNDS_SMALL = 40; // ranges up to this many bins go to the small-footprint launch

__global__ __launch_bounds__(256) void k_bounds_check(DeviceState st, int ktype, int margin,
                                                      int32_t *scratch)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * blockDim.x + threadIdx.x; // one lane per cell
    const int B = st.n_bin, G = B * st.n_class;
    int need = -1;                                        // -1 none, 0 small range, 1 large
    if (c < st.n_cells) {
        int plo = B, phi = -1;
        for (int g = 0; g < G; g++)
            if (st.gcnt[(int64_t)c * G + g] > 0) {
                const int b = g % B;
                plo = b < plo ? b : plo;
                phi = b > phi ? b : phi;
            }
        int32_t *rng = st.kmax_rng + 4 * (int64_t)c;
        const double temp = st.temp[c], pres = st.pressure[c];
        const int reach = phi + 2 < B - 1 ? phi + 2 : B - 1;
        if (!(rng[2] == ktype && st.kmax_env[2 * (int64_t)c] == temp &&
              st.kmax_env[2 * (int64_t)c + 1] == pres && plo >= rng[0] && reach <= rng[1])) {
            rng[2] = 0;                                   // void until the table is written
            if (phi < 0) { rng[0] = 0; rng[1] = -1; }     // empty cell: nothing to bound
            else { rng[0] = plo; rng[1] = phi + margin < B - 1 ? phi + margin : B - 1; }
            need = (rng[1] - rng[0] + 1 > BOUNDS_SMALL) ? 1 : 0;
        }
    }
    // scratch: {n_small, n_large, small list [n_cells], large list [n_cells]}; one atomic per
    // wavefront and list
    for (int big = 0; big < 2; big++) {
        const uint64_t m = __ballot(need == big);
        if (m == 0) continue;
        int base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&scratch[big], (int)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m), 64);
        if (need == big)
            scratch[2 + (int64_t)big * st.n_cells + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = c;
    }
}

// list == nullptr: the whole table of every cell.  Otherwise the listed cells, each over the
// range k_bounds_check left in kmax_rng (at most bins_cap bins: the LDS tables are laid out
// for that many).  The workgroups share the work round-robin.
__global__ __launch_bounds__(512) void k_kernel_bounds(DeviceState st, int ktype, int bins_cap,
                                                       const int32_t *list, const int32_t *count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = st.n_bin, L = 2 * bins_cap + 1;
    const int ND = (st.rho_min == st.rho_max) ? 1 : 2;
    const int n_work = list ? *count : st.n_cells;
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int c = list ? list[w] : w;
    const double temp = st.temp[c], pres = st.pressure[c];
    int klo = 0, khi = B - 1;
    if (list) { klo = st.kmax_rng[4 * (int64_t)c]; khi = st.kmax_rng[4 * (int64_t)c + 1]; }
    const int nb = khi - klo + 1;                 // bins, lattice points 2 klo .. 2 khi + 2
    const int Ls = nb > 0 ? 2 * nb + 1 : 0, l0 = 2 * klo;
    const int NP = Ls * ND;
    // per-(lattice point, density) particle properties, structure of arrays in LDS
    double *p_rad = (double *)smem, *p_dif = p_rad + 2 * L, *p_spd = p_dif + 2 * L,
           *p_del = p_spd + 2 * L, *p_fal = p_del + 2 * L;
    double *p_vol = p_fal + 2 * L;          // [L] the lattice volumes
    float *M = (float *)(p_vol + L + 1);    // row l1 of the upper triangle: Ls - l1 entries
    const EnvDerived env = env_derive(temp, pres);
    const KernelParams kp{st.const_k, st.additive_beta};
    for (int j = threadIdx.x; j < NP; j += blockDim.x) {
        const int l = j / ND, d = j - l * ND, lg = l0 + l;
        const double v = (lg & 1) ? st.v_samp[3 * (lg >> 1) + 1] : st.v_edge[lg >> 1];
        if (d == 0) p_vol[l] = v;
        const double rho = d ? st.rho_max : st.rho_min;
        const PartProps q = props_for(ktype, v, v * rho, env);
        p_rad[j] = q.rad; p_dif[j] = q.diffus; p_spd[j] = q.speedsq; p_del[j] = q.deltasq;
        p_fal[j] = q.wfall;
    }
    __syncthreads();
    const int n_lat = Ls * (Ls + 1) / 2;
    // a thread walks the triangle in steps of blockDim.x: (l1, position in row l1) kept incrementally
    int l1 = 0, pos = threadIdx.x;
    for (int j = threadIdx.x; j < n_lat; j += blockDim.x, pos += blockDim.x) {
        while (pos >= Ls - l1) { pos -= Ls - l1; l1++; }
        const int l2 = l1 + pos;
        const double v1 = p_vol[l1], v2 = p_vol[l2];
        double m = 0.0;
        for (int d1 = 0; d1 < ND; d1++) {
            const int ja = l1 * ND + d1;
            const PartProps qa{p_rad[ja], p_dif[ja], p_spd[ja], p_del[ja], p_fal[ja]};
            for (int d2 = 0; d2 < ND; d2++) {
                const int jb = l2 * ND + d2;
                const PartProps qb{p_rad[jb], p_dif[jb], p_spd[jb], p_del[jb], p_fal[jb]};
                const double k = kernel_pair(ktype, v1, v2, qa, qb, kp);
                if (k > m) m = k;
            }
        }
        // smallest float >= m: half the table, four bounds per 16-byte load in the sampler
        float f = (float)m;
        if ((double)f < m) f = __uint_as_float(__float_as_uint(f) + 1u);
        M[j] = f;
    }
    __syncthreads();
    // the entries of the range (both triangles: row segments are written whole, coalesced)
    float *km = st.kmax + (int64_t)c * st.kmax_stride;
    for (int j = threadIdx.x; j < nb * nb; j += blockDim.x) {
        const int r1 = j / nb, r2 = j - r1 * nb;
        float f = 0.0f;
        for (int s1 = 0; s1 < 3; s1++)
            for (int s2 = 0; s2 < 3; s2++) {
                int la = 2 * r1 + s1, lb = 2 * r2 + s2;
                if (la > lb) { const int x = la; la = lb; lb = x; }
                const float k = M[la * (2 * Ls + 1 - la) / 2 + (lb - la)];
                if (k > f) f = k;
            }
        km[(klo + r1) * B + (klo + r2)] = f;
    }
    if (threadIdx.x == 0) {
        int32_t *rng = st.kmax_rng + 4 * (int64_t)c;
        rng[0] = klo; rng[1] = khi; rng[2] = ktype;
        st.kmax_env[2 * (int64_t)c] = temp; st.kmax_env[2 * (int64_t)c + 1] = pres;
    }
    __syncthreads(); // the LDS tables are reused by the next cell
  }
}

// ---------------------------------------------------------------- bin sort
//
// aero_state_sort [upstream; called at wrf_pmc_init.F90:1279 and inside
// mc_coag]: groups g = class*n_bin + bin, ascending slot order inside a
// group.  One 256-thread workgroup per cell.  Each wavefront owns a
// contiguous quarter of the slots; per 64-slot chunk the lanes that share a
// group are found with ballots (rank = popcount of lower matching lanes), so
// the order is the slot order by construction and never depends on atomics
// or scheduling.  vtot is streamed once (8 B per particle), perm written
// once (4 B per particle).  The lists are laid out one after the other in
// perm with sort_slack() free entries behind each, so that the sampler can
// append in place; a cell whose lists still hold (sort_valid, the reference's
// aero_state%valid_sort, wrf_pmc_trans_aero.F90:2094) is skipped.

// Free entries left behind a list of n when the lists are laid out (oracle: pmco_sort_slack): 16 + n/8
// behind every group, the empty ones too, while the cell's perm has the entries for that (level 0:
// every block of up to 512 groups whose cap leaves the offsets 16 bits).  With tens of weight
// classes (registry.partmc_trans:2-41: 40 classes x 80 bins = 3200 groups, nearly all empty) or a
// cell near 65 280 slots perm holds less -- pcap = min(65535, cap + cap/8 + min(16 G, 8192)) -- and
// the room shrinks, first behind the empty groups, by the first level of a fixed ladder that fits
// the cell's N particles, P populated groups and S3 = sum n_g / 8:
//   level   populated       empty
//     0     16 + n/8         16
//     1     16 + n/8          4
//     2     16 + n/8          1
//     3     16 + n/8          0
//     4      4 + n/8          0
//     5          n/8          0
//     6          0            0       (N <= cap <= pcap: always fits)
// An entry that finds no room goes to the table of appends (max_appends), as before.
__host__ __device__ inline uint32_t sort_slack_level(uint32_t n, int level)
{
    if (n == 0) return level == 0 ? 16u : level == 1 ? 4u : level == 2 ? 1u : 0u;
    return (level <= 3 ? 16u : level == 4 ? 4u : 0u) + (level <= 5 ? (n >> 3) : 0u);
}
__host__ __device__ inline int sort_layout_level(uint32_t N, uint32_t S3, uint32_t P, uint32_t G, uint32_t pcap)
{
    const uint32_t base[7] = {16u, 16u, 16u, 16u, 4u, 0u, 0u}, empty[7] = {16u, 4u, 1u, 0u, 0u, 0u, 0u};
    for (int l = 0; l < 6; l++)
        if ((uint64_t)N + S3 + (uint64_t)base[l] * P + (uint64_t)empty[l] * (G - P) <= pcap) return l;
    return 6;
}

// exclusive scan in place over an LDS u32 array of length n
__device__ __forceinline__ void block_excl_scan_u32(uint32_t *a, int n, uint32_t *wsum)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    const int j0 = min(n, t * per), j1 = min(n, j0 + per);
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

// Which cells have work to do in the kernel that follows: most calls find most cells idle (the
// lists of the last call still hold, no cell is past its tombstone threshold, no class is off
// its ideal count), and a workgroup launched only to find that out costs more than the look.
// One lane per cell decides; the cells with work are listed (scratch[0] = count, the list
// behind it; one atomic per wavefront) and the kernel's workgroups share the list round-robin.
enum : int { LIST_SORT = 0, LIST_COMPACT = 1, LIST_REBALANCE = 2 };

__global__ __launch_bounds__(256) void k_list_cells(DeviceState st, int what, int p0, int p1,
                                                    int cell_first, int n_do, int32_t *scratch)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = cell_first + k;
    bool need = false;
    if (k < n_do) {
        if (what == LIST_SORT) {
            need = st.sort_valid[c] == 0;
        } else if (what == LIST_COMPACT) { // more than p0 / p1 of the slots are tombstones
            const int n_hi = st.n_hi[c], n_live = st.n_part[c];
            need = (int64_t)(n_hi - n_live) * p1 > (int64_t)n_hi * p0;
        } else { // one class: its count is the cell's; p0 = allow_doubling, p1 = allow_halving
            const double n = (double)st.n_part[c], ideal = st.n_ideal[c];
            need = (p0 && n > 0.0 && n < ideal / 2.0) || (p1 && n > ideal * 2.0);
        }
    }
    const uint64_t m = __ballot(need);
    if (m == 0) return;
    int base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(&scratch[0], (int)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    if (need) scratch[1 + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = c;
}

// stage: the cell's keys are copied to LDS (2 bytes a slot of `cap`) and both passes read them there; 0: they
// are read from memory twice, the second time 512 at a time through a small buffer of the wavefront's own
// (the default: with ten classes and cap = 22 016 the copy left one workgroup a CU where three fit
// without it, and with one class the form without the copy is the faster one too)
__global__ __launch_bounds__(256) void k_bin_sort(DeviceState st, const int32_t *list,
                                                  const int32_t *count, int stage)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_lay[4]; // the layout's totals: particles, sum n_g / 8, populated groups; its level
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // wavefronts of the workgroup (4; 2 or 1 where the per-wavefront tables of 4 would not fit in
    // LDS: tens of weight classes), each with a contiguous share of the slots
    const int NW = (int)blockDim.x >> 6, NT = (int)blockDim.x;
    const int B = st.n_bin, C = st.n_class, G = B * C;
    const int n_work = list ? *count : st.n_cells;
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int c = list ? list[w] : w;
    if (st.sort_valid[c]) continue; // the lists of the last call still hold (whole workgroup)
    const int n = st.n_hi[c]; // slots in use; dead ones (vtot < 0) are skipped
    const int64_t cb = (int64_t)c * st.cap;

    const size_t o_tab = (sizeof(double) * (size_t)(B + 1) + 15) & ~(size_t)15;
    const size_t o_key = (o_tab + (sizeof(uint64_t) * NW + sizeof(uint32_t) * (NW + 1)) * (size_t)G + 15) & ~(size_t)15;
    double *edge = (double *)smem;                    // [B+1]
    uint64_t *pmask = (uint64_t *)(smem + o_tab);     // [NW][G] lanes of the chunk holding group g
    uint32_t *hist = (uint32_t *)(pmask + NW * (size_t)G); // [NW][G] per-wave counts -> cursors
    uint32_t *tot = hist + NW * (size_t)G;            // [G] totals -> offsets
    uint16_t *key = (uint16_t *)(smem + o_key);       // [cap] group of slot (16-byte aligned)

    for (int j = t; j <= B; j += NT) edge[j] = st.v_edge[j];
    for (int j = t; j < NW * G; j += NT) { hist[j] = 0; pmask[j] = 0; }
    if (t < 4) s_lay[t] = 0;
    const bool keyed = st.key_valid[c] != 0; // (the whole workgroup: the slots' keys still hold)
    uint16_t *gk = st.gkey + cb;
    __syncthreads();

    // this wave's slots: a contiguous range whose length is a multiple of 64
    const int per = (((n + NW - 1) / NW) + 63) & ~63;
    const int lo = min(n, wave * per), hi = min(n, lo + per);
    const Hot *vt = st.hot + cb; // .vtot = cached total volume
    const int32_t *wc = st.wclass + cb;
    const double v_lo = edge[0], v_hi = edge[B];
    uint32_t *myhist = hist + (size_t)wave * G;

    const uint64_t lt = (1ull << lane) - 1ull;

    int oog = 0;
    if (keyed) {
        // pass 1, keys kept: 2 bytes per slot instead of its 32-byte record.  The slots the pair
        // sampler re-binned or removed since the keys were written are in the cell's event log,
        // applied in order (a slot may appear more than once) and written back to the keys.
        if (!stage) {
            // the log's entries go to the keys in memory: of several for one slot the last one
            const int nl = min(st.n_log[c], EVLOG_CAP);
            __shared__ uint32_t s_log2[EVLOG_CAP];
            if (nl > 0) {
                for (int j = t; j < nl; j += NT) s_log2[j] = st.evlog[(int64_t)c * EVLOG_CAP + j];
                __syncthreads();
                for (int j = t; j < nl; j += NT) {
                    const uint32_t e = s_log2[j], sl = e & 0xFFFFu;
                    bool last = (int)sl < n;
                    for (int q = j + 1; q < nl && last; q++) last = (s_log2[q] & 0xFFFFu) != sl;
                    if (last) gk[sl] = (uint16_t)(e >> 16);
                }
                __threadfence_block();
                __syncthreads();
            }
            if ((st.cap & 7) == 0) {
                // eight keys per lane and request (the range starts at a multiple of 64 slots, the cell's keys
                // at a multiple of 8; keys past hi, below cap, ride along and are not counted)
                const uint4 *g4 = (const uint4 *)(gk + lo);
                const int n8 = (hi - lo + 7) >> 3;
                for (int j = lane; j < n8; j += 64) {
                    const uint4 v = g4[j];
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int h = 0; h < 8; h++) {
                        const uint32_t g = (w4[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                        if (g != 0xFFFFu && lo + j * 8 + h < hi) atomicAdd(&myhist[g], 1u);
                    }
                }
            } else
            for (int i0 = lo; i0 < hi; i0 += 256) {
                uint32_t g4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) g4[u] = i0 + u * 64 + lane < hi ? (uint32_t)gk[i0 + u * 64 + lane] : 0xFFFFu;
#pragma unroll
                for (int u = 0; u < 4; u++) if (g4[u] != 0xFFFFu) atomicAdd(&myhist[g4[u]], 1u);
            }
        } else {
        if ((st.cap & 7) == 0) {
            // eight keys per lane and request: the wave's whole range is in flight at once.  The
            // range starts at a multiple of 64 slots and the cell's keys at a multiple of 8: both
            // the global and the LDS side are 16-byte aligned; keys past hi (below cap) ride along
            // and are never looked at.
            const uint4 *g4 = (const uint4 *)(gk + lo);
            uint4 *k4 = (uint4 *)(key + lo);
            const int n8 = (hi - lo + 7) >> 3;
            for (int j = lane; j < n8; j += 64) k4[j] = g4[j];
        } else {
            for (int i = lo + lane; i < hi; i += 64) key[i] = gk[i];
        }
        __syncthreads();
        const int nl = min(st.n_log[c], EVLOG_CAP);
        if (nl > 0) {
            // (the log is read by all threads at once -- one entry each, EVLOG_CAP <= 256 -- and
            // applied from LDS by one: a later entry of the same slot must win)
            __shared__ uint32_t s_log[EVLOG_CAP];
            for (int j = t; j < nl; j += NT) s_log[j] = st.evlog[(int64_t)c * EVLOG_CAP + j];
            __syncthreads();
            if (t == 0)
                for (int j = 0; j < nl; j++) {
                    const uint32_t e = s_log[j];
                    if ((int)(e & 0xFFFFu) < n) key[e & 0xFFFFu] = (uint16_t)(e >> 16);
                }
            __syncthreads();
            for (int j = t; j < nl; j += NT) {
                const uint32_t sl = s_log[j] & 0xFFFFu;
                if ((int)sl < n) gk[sl] = key[sl];
            }
        }
        for (int i = lo + lane; i < hi; i += 64) {
            const uint32_t g = key[i];
            if (g != 0xFFFFu) atomicAdd(&myhist[g], 1u);
        }
        }
    } else {
    // pass 1: keys + per-wave histogram (LDS atomics: counting is order-independent).
    // Four loads are issued before the first use so a wave keeps 2 KiB in flight.
    for (int i0 = lo; i0 < hi; i0 += 256) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            v[u] = i < hi ? vt[i].vtot : v_lo;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            if (i < hi) {
                if (v[u] >= 0.0) {
                    oog |= (v[u] < v_lo) | (v[u] >= v_hi);
                    uint32_t g = (uint32_t)bin_index(edge, B, v[u], st);
                    if (C > 1) g += (uint32_t)wc[i] * (uint32_t)B;
                    if (stage) key[i] = (uint16_t)g;
                    gk[i] = (uint16_t)g;
                    atomicAdd(&myhist[g], 1u);
                } else {
                    if (stage) key[i] = 0xFFFFu; // tombstone
                    gk[i] = (uint16_t)0xFFFFu;
                }
            }
        }
    }
    }
    if (oog) atomicOr(&st.status[c], ST_OUT_OF_GRID);
    if (!stage) __threadfence_block(); // (the keys written to memory are read again below, by other lanes)
    __syncthreads();

    int32_t *gcnt_g = st.gcnt + (int64_t)c * G;
    int32_t *goff_g = st.goff + (int64_t)c * (G + 1);
    {
        uint32_t lN = 0, lS3 = 0, lP = 0;
        for (int g = t; g < G; g += NT) {
            uint32_t s = 0;
            for (int q = 0; q < NW; q++) { const uint32_t h = hist[q * G + g]; hist[q * G + g] = s; s += h; }
            tot[g] = s;
            gcnt_g[g] = (int32_t)s;
            lN += s; lS3 += s >> 3; lP += s > 0;
        }
        // (the layout's level: one of a fixed ladder, by the cell's totals -- sort_slack_level)
        for (int d = 32; d >= 1; d >>= 1) {
            lN += __shfl_xor(lN, d, 64); lS3 += __shfl_xor(lS3, d, 64); lP += __shfl_xor(lP, d, 64);
        }
        if (lane == 0) { atomicAdd(&s_lay[0], lN); atomicAdd(&s_lay[1], lS3); atomicAdd(&s_lay[2], lP); }
    }
    __syncthreads();
    const int level = sort_layout_level(s_lay[0], s_lay[1], s_lay[2], (uint32_t)G, (uint32_t)st.pcap);
    for (int g = t; g < G; g += NT) tot[g] = tot[g] + sort_slack_level(tot[g], level); // the list and the room behind it
    __syncthreads();
    block_excl_scan_u32(tot, G, wsum);
    for (int g = t; g < G; g += NT) {
        const uint32_t o = tot[g];
        goff_g[g] = (int32_t)o;
        if (g == G - 1) {
            const uint32_t s = (uint32_t)gcnt_g[g];
            goff_g[G] = (int32_t)(o + s + sort_slack_level(s, level));
        }
        for (int q = 0; q < NW; q++) hist[q * G + g] += o;
    }
    __syncthreads();

    // scatter: position = cursor of (wave, group) + rank among the chunk's lanes of the
    // same group.  The lanes of a group are collected by OR-ing their lane bit into the
    // group's 64-bit word in LDS (commutative, so the result does not depend on the order
    // the hardware serves the atomics); rank = popcount of the lower lanes in that word.
    uint16_t *perm = st.perm + (int64_t)c * st.pcap;
    uint64_t *mymask = pmask + (size_t)wave * G;
    if (t == 0) { st.sort_valid[c] = 1; st.key_valid[c] = 1; st.n_log[c] = 0; }
    // without the whole copy: 512 keys at a time through a small LDS buffer of the wavefront's own (two
    // halves: the block being scattered, the next one's request in flight meanwhile), eight keys a lane
    const bool rolling = !stage && (st.cap & 7) == 0;
    uint16_t *roll = key + (size_t)wave * 1024;
    const uint4 *gk4 = (const uint4 *)(st.gkey + cb);
    uint4 nxt = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (rolling && lo + lane * 8 < hi) nxt = gk4[(lo >> 3) + lane];
    uint32_t k_next = (!stage && !rolling && lo + lane < hi) ? (uint32_t)gk[lo + lane] : 0xFFFFu; // (requested a chunk ahead)
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        uint32_t gk;
        if (stage) {
            gk = i < hi ? key[i] : 0xFFFFu;
        } else if (rolling) {
            const int rel = i0 - lo, half = (rel >> 9) & 1;
            if ((rel & 511) == 0) {   // (uniform) a new block: what was requested goes to LDS, the next block is requested
                ((uint4 *)(roll + half * 512))[lane] = nxt;
                nxt = (i0 + 512 + lane * 8 < hi) ? gk4[((i0 + 512) >> 3) + lane] : make_uint4(~0u, ~0u, ~0u, ~0u);
                lds_sync();
            }
            gk = i < hi ? (uint32_t)roll[half * 512 + (rel & 511) + lane] : 0xFFFFu;
        } else {
            gk = k_next;
            k_next = i + 64 < hi ? (uint32_t)(st.gkey + cb)[i + 64] : 0xFFFFu;
        }
        const bool valid = gk != 0xFFFFu;
        if (valid) atomicOr((unsigned long long *)&mymask[gk], 1ull << lane);
        lds_sync();
        uint64_t peers = 0;
        uint32_t cur = 0;
        if (valid) { peers = mymask[gk]; cur = myhist[gk]; }
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        lds_sync(); // every lane has read the word and the cursor
        if (valid) perm[cur + rank] = (uint16_t)i; // (not waited for: nothing here reads perm)
        if (valid && rank == 0) { // the group's first lane advances the cursor, clears the word
            myhist[gk] = cur + (uint32_t)__popcll(peers);
            mymask[gk] = 0;
        }
        lds_sync();
    }
    __syncthreads(); // the LDS tables are reused by the next cell
  }
}

// ------------------------------------------------------------- coag sampler
//
// mc_coag sweep.  One group of L lanes (L = 16: four cells per wavefront;
// L = 64: one) owns a cell.  The reference's sequential semantics (classes,
// bins, samples in order; every accepted pair changes the lists the next
// sample draws from) are kept exactly: the group evaluates up to L upcoming
// candidates of the current row against the current lists, the first accepted
// one (ballot + ctz) is applied, and the Poisson counts of the row's later
// slots are redrawn from the new counts.  Draws are addressed by
// (seed, cell, step, slot, sample), so a redraw with unchanged counts
// reproduces the same number.  All groups of a wavefront run one flattened
// loop (advance row | draw counts | prefix | evaluate | coagulate) so they
// stay convergent whatever row each one is on.

struct CellLds {
    EnvDerived *env;    // per-cell air properties (kept out of registers)
    uint64_t *mask;     // [n_class][2] non-empty bins bitmask
    uint32_t *P;        // [WN] inclusive prefix of remaining samples of the window's slots
    uint32_t *ns;       // [WN] Poisson counts of the window's slots
    uint16_t *cnt;      // [G] current list length
    uint16_t *nb;       // [G] entries still in the sorted base segment
    uint16_t *off;      // [G] base segment offset in perm
    uint16_t *room;     // [G] entries the base segment can hold
    uint16_t *last_app; // [G] newest overflow entry of the group (0xFFFF none)
    uint16_t *app_slot; // [max_appends] overflow entries: appended when the room was used up
    uint16_t *app_prev; // [max_appends]
    uint8_t *act;       // [WN] partner bins of the window's slots
};

__host__ __device__ inline size_t sampler_lds_bytes(int G, int C, int max_appends, int wn)
{
    size_t o = 0;
    o += sizeof(EnvDerived);                // env
    o += sizeof(uint64_t) * 2 * (size_t)C;  // mask
    o += sizeof(uint32_t) * 2 * (size_t)wn; // P, ns
    o += sizeof(uint16_t) * 5 * (size_t)G;  // cnt nb off room last_app
    o += sizeof(uint16_t) * 2 * (size_t)max_appends;
    o += (size_t)wn;                        // act
    return (o + 15) & ~(size_t)15;
}

struct CellCtx {
    CellLds L;
    uint16_t *perm;
    Hot *hot; // (cached total volume, cached mass, create times) per slot
};

__device__ __forceinline__ uint32_t list_get(const CellCtx &X, int g, int i)
{
    const int nb = X.L.nb[g];
    if (i < nb) return ld_u16(&X.perm[X.L.off[g] + i]);
    int e = X.L.last_app[g];
    for (int k = X.L.cnt[g] - 1 - i; k > 0; k--) e = X.L.app_prev[e];
    return X.L.app_slot[e];
}

// group-uniform: every lane of the group performs the same LDS writes
__device__ __forceinline__ void list_set(const CellCtx &X, int g, int i, uint32_t val, bool lead)
{
    const int nb = X.L.nb[g];
    if (i < nb) {
        if (lead) st_u16(&X.perm[X.L.off[g] + i], val);
    } else {
        int e = X.L.last_app[g];
        for (int k = X.L.cnt[g] - 1 - i; k > 0; k--) e = X.L.app_prev[e];
        X.L.app_slot[e] = (uint16_t)val;
    }
}

// swap-with-last removal of logical position i; returns the old last index
__device__ __forceinline__ int list_remove(const CellCtx &X, int g, int i, int B, bool lead)
{
    const int last = (int)X.L.cnt[g] - 1;
    if (i != last) list_set(X, g, i, list_get(X, g, last), lead);
    if (last >= (int)X.L.nb[g]) X.L.last_app[g] = X.L.app_prev[X.L.last_app[g]];
    else X.L.nb[g] = (uint16_t)last;
    X.L.cnt[g] = (uint16_t)last;
    if (last == 0) {
        const int cl = g / B, b = g - cl * B;
        X.L.mask[cl * 2 + (b >> 6)] &= ~(1ull << (b & 63));
    }
    return last;
}

// Appends behind the list in perm while there is room (and nothing of the group sits in
// the overflow list already), else to overflow entry e.  Returns 1 when e was used.
__device__ __forceinline__ int list_append(const CellCtx &X, int g, uint32_t slot, int e, int B,
                                           bool lead)
{
    const int n = X.L.cnt[g];
    int used = 0;
    if (X.L.last_app[g] == 0xFFFF && n < (int)X.L.room[g]) {
        if (lead) st_u16(&X.perm[X.L.off[g] + n], slot);
        X.L.nb[g] = (uint16_t)(n + 1);
    } else {
        X.L.app_slot[e] = (uint16_t)slot;
        X.L.app_prev[e] = X.L.last_app[g];
        X.L.last_app[g] = (uint16_t)e;
        used = 1;
    }
    X.L.cnt[g] = (uint16_t)(n + 1);
    if (n == 0) {
        const int cl = g / B, b = g - cl * B;
        X.L.mask[cl * 2 + (b >> 6)] |= (1ull << (b & 63));
    }
    return used;
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

template <int L>
struct Grp {
    // ballot restricted to this lane's group, as the low L bits
    static __device__ __forceinline__ uint64_t ballot(bool p, int sub)
    {
        const uint64_t m = __ballot(p);
        if (L == 64) return m;
        return (m >> (sub * L)) & ((1ull << (L & 63)) - 1ull);
    }
    static __device__ __forceinline__ uint32_t incl_scan(uint32_t x, int gl)
    {
#pragma unroll
        for (int d = 1; d < L; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, L);
            if (gl >= d) x += y;
        }
        return x;
    }
    static __device__ __forceinline__ uint32_t sum(uint32_t x)
    {
#pragma unroll
        for (int d = L / 2; d >= 1; d >>= 1) x += __shfl_xor(x, d, L);
        return x;
    }
};

// launch bounds: 3 wavefronts per SIMD (<= 168 VGPRs) measured fastest; 4 spills, 2 idles
template <int L>
__global__ __launch_bounds__(256, 3) void k_coag_sample(DeviceState st, int ktype, double del_t,
                                                     uint64_t seed, uint32_t step,
                                                     uint32_t lds_per_cell)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int GPW = 64 / L; // cells per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / L, gl = lane % L;
    const bool lead = gl == 0;
    const int cpb = (blockDim.x >> 6) * GPW;
    const int cidx = blockIdx.x * cpb + wave * GPW + sub;
    const bool in_range = cidx < st.n_cells;
    const bool live = in_range && in_loop(st, cidx);
    const int c = live ? cidx : st.n_cells - 1;
    const int B = st.n_bin, C = st.n_class, G = B * C;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << gl) - 1ull;
    if (in_range && !live && lead) { st.n_samp[cidx] = 0; st.n_coag[cidx] = 0; st.n_events[cidx] = 0; }

    // sort-grid edges, shared by the workgroup
    double *edge = (double *)smem;
    for (int j = threadIdx.x; j <= B; j += blockDim.x) edge[j] = st.v_edge[j];
    __syncthreads(); // the only workgroup barrier: groups finish at different times

    CellCtx X;
    {
        unsigned char *base = smem + (((size_t)(B + 1) * sizeof(double) + 15) & ~(size_t)15) +
                              (size_t)lds_per_cell * (size_t)(wave * GPW + sub);
        size_t o = 0;
        constexpr int WNL = (L == 64) ? 64 : PMCX_WIN;
        X.L.env = (EnvDerived *)(base + o); o += sizeof(EnvDerived);
        X.L.mask = (uint64_t *)(base + o); o += sizeof(uint64_t) * 2 * (size_t)C;
        X.L.P = (uint32_t *)(base + o); o += sizeof(uint32_t) * (size_t)WNL;
        X.L.ns = (uint32_t *)(base + o); o += sizeof(uint32_t) * (size_t)WNL;
        X.L.cnt = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.nb = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.off = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.room = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.last_app = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)G;
        X.L.app_slot = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)st.max_appends;
        X.L.app_prev = (uint16_t *)(base + o); o += sizeof(uint16_t) * (size_t)st.max_appends;
        X.L.act = (uint8_t *)(base + o);
    }
    X.perm = st.perm + (int64_t)c * st.pcap;
    X.hot = st.hot + cb;

    int32_t *gcnt_g = st.gcnt + (int64_t)c * G;
    const int32_t *goff_g = st.goff + (int64_t)c * (G + 1);
    for (int g = gl; g < G; g += L) {
        const int n = gcnt_g[g];
        X.L.cnt[g] = (uint16_t)n;
        X.L.nb[g] = (uint16_t)n;
        X.L.off[g] = (uint16_t)goff_g[g];
        X.L.room[g] = (uint16_t)(goff_g[g + 1] - goff_g[g]);
        X.L.last_app[g] = 0xFFFF;
    }
    for (int j = gl; j < 2 * C; j += L) X.L.mask[j] = 0;
    wave_sync();
    for (int cl = 0; cl < C; cl++)
        for (int b0 = 0; b0 < B; b0 += L) {
            const int b = b0 + gl;
            const uint64_t m = Grp<L>::ballot(b < B && gcnt_g[cl * B + (b < B ? b : 0)] > 0, sub);
            // bins b0..b0+L-1 never straddle a 64-bit word (L divides 64)
            X.L.mask[cl * 2 + (b0 >> 6)] |= m << (b0 & 63);
            wave_sync();
        }

    *X.L.env = env_derive(st.temp[c], st.pressure[c]); // same value from every lane of the group
    wave_sync();
    const EnvDerived &env = *X.L.env;
    const KernelParams kp{st.const_k, st.additive_beta};
    const float *kmax = st.kmax + (int64_t)c * st.kmax_stride;
    const double *mag = st.mag + (int64_t)c * C;
    int n_ev = 0, n_app = 0, n_dead = 0, tot_coag = 0, status = 0;
    uint64_t tot_samp = 0;
    // the cell's event log for the next sort (as in k_coag_lane): uniform within the group
    int n_lg = st.key_valid[c] != 0 ? st.n_log[c] : -1;

    // Row state (uniform inside the group).  The partner slots of a row are
    // visited through a window of up to WN non-empty partner bins: counts are
    // drawn for the window only, and the window is rebuilt from the slot of an
    // event (whose own count and progress are kept) or moved on when exhausted.
    constexpr int WN = (L == 64) ? 64 : PMCX_WIN;
    int c1 = 0, c2 = 0, b1 = -1;
    int gA = 0, cc = 0, wn = 0, counted = 0, wfrom = 0;
    uint32_t cur_s0 = 0, ns0 = 0;
    double n1 = 0, n2 = 0, nmin = 0, fmax = 0, factor = 0;
    uint32_t T = 0, q0 = 0; // a window holds < 2^31 samples (NSAMP_MAX per slot)
    bool need_row = true, need_win = false, keep_first = false;
    bool done = !live;

    auto sum_ns = [&](int from, int to) { // sum of ns[from..to)
        uint32_t s = 0;
        for (int j = from + gl; j < to; j += L) s += X.L.ns[j];
        return (uint64_t)Grp<L>::sum(s);
    };

    while (true) {
        // ---- advance to the next non-empty row (c1, c2, b1) ----
        if (!done && need_row) {
            while (true) {
                b1 = next_bit(&X.L.mask[c1 * 2], b1 + 1);
                if (b1 >= 0 && b1 < B) break;
                b1 = -1;
                if (++c2 > c1) { c2 = 0; ++c1; }
                if (c1 >= C) { done = true; break; }
            }
            if (!done) {
                n1 = mag[c1]; n2 = mag[c2];
                cc = (n1 < n2) ? c1 : c2;
                nmin = (n1 < n2) ? n1 : n2;
                fmax = (n1 < n2) ? n2 : n1;
                factor = n1 * n2 / nmin;
                gA = c1 * B + b1;
                wfrom = (c1 == c2) ? b1 : 0;
                keep_first = false; cur_s0 = 0; counted = 0;
                need_win = true; need_row = false;
            }
        }
        if (__ballot(!done) == 0) break;
        if (done) continue;

        // ---- (re)build the window: bins, Poisson counts, prefix ----
        if (need_win) {
            // slot 0 is the slot of the last event when keep_first (its bin is wfrom - 1)
            int n = 0;
            if (keep_first) { X.L.act[0] = (uint8_t)(wfrom - 1); n = 1; }
            for (int b0 = (wfrom / L) * L; b0 < B && n < WN; b0 += L) {
                const int b = b0 + gl;
                const bool bit = b < B && b >= wfrom &&
                                 ((X.L.mask[c2 * 2 + (b >> 6)] >> (b & 63)) & 1ull);
                const uint64_t m = Grp<L>::ballot(bit, sub);
                const int pos = n + __popcll(m & lt);
                if (bit && pos < WN) X.L.act[pos] = (uint8_t)b;
                n += __popcll(m);
            }
            wn = n < WN ? n : WN;
            wave_sync();
            if (wn == 0) { need_row = true; need_win = false; continue; }

            const int nA = X.L.cnt[gA];
            uint32_t carry = 0;
            for (int j0 = 0; j0 < wn; j0 += L) {
                const int j = j0 + gl;
                uint32_t ns = 0, r = 0;
                if (j < wn) {
                    if (keep_first && j == 0) {
                        ns = ns0;
                        r = ns0 - cur_s0;
                    } else {
                        const int b2 = X.L.act[j];
                        const int gB = c2 * B + b2;
                        const int nB = X.L.cnt[gB];
                        const bool same = (gA == gB);
                        if (nA > 0 && nB > 0 && !(same && nA < 2)) {
                            const double k_max = kmax[b1 * B + b2] * fmax;
                            const double n_possible =
                                same ? (double)nA * ((double)nA - 1.0) / 2.0
                                     : (double)nA * (double)nB;
                            const double mean = k_max * del_t * n_possible;
                            // (u < 1 - mean - 4e-15 implies u < det_exp(-mean): a count of zero, as poisson()
                            // says, without the exponential -- nearly every slot of a sweep, and with tens of
                            // weight classes there are thousands of them a cell)
                            const double u_slot = poisson_uniform(seed, cell_id, step, (uint32_t)gA, (uint32_t)gB);
                            if (!(u_slot < (1.0 - mean) - 4e-15)) ns = (uint32_t)poisson(mean, u_slot);
                        }
                        r = ns;
                    }
                    X.L.ns[j] = ns;
                }
                const uint32_t inc = Grp<L>::incl_scan(r, gl) + carry;
                if (j < wn) X.L.P[j] = inc;
                carry = __shfl(inc, L - 1, L);
            }
            wave_sync();
            T = carry;
            q0 = 0;
            counted = keep_first ? 1 : 0;
            need_win = false;
        }

        // ---- window exhausted: its slots have all started; move on ----
        if (q0 >= T) {
            tot_samp += sum_ns(counted, wn);
            wfrom = (int)X.L.act[wn - 1] + 1;
            keep_first = false; cur_s0 = 0;
            // a window that was not filled held every remaining bin of the row
            if (wfrom >= B || wn < WN) need_row = true; else need_win = true;
            continue;
        }

        // ---- evaluate up to L candidates against the current lists ----
        // Stores of the previous event (lists, caches, merged species) must have reached
        // L2 before the gathers below; waiting here instead of right after the event lets
        // the window rebuild above run under their latency.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        const uint32_t q = q0 + (uint32_t)gl;
        const bool act = q < T;
        int j = 0;
        uint32_t s = 0;
        int i1 = 0, i2 = 0;
        uint32_t p1 = 0, p2 = 0;
        double v1 = 0, v2 = 0, m1 = 0, m2 = 0;
        bool acc = false, over = false;
        if (act) {
            int lo = 0, hi = wn - 1; // smallest j with P[j] > q
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (X.L.P[mid] > q) hi = mid; else lo = mid + 1;
            }
            j = lo;
            const int b2 = X.L.act[j];
            const uint32_t first = (keep_first && j == 0) ? cur_s0 : 0u;
            const uint32_t rem = X.L.ns[j] - first;
            s = (q - (X.L.P[j] - rem)) + first;
            const int gB = c2 * B + b2;
            const bool same = (gA == gB);
            const int nA = X.L.cnt[gA], nB = X.L.cnt[gB];
            uint32_t w[4];
            rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB, SITE_CAND, s, w);
            i1 = (int)index_of(w[0], (uint32_t)nA);
            if (same) {
                i2 = (int)index_of(w[1], (uint32_t)(nA - 1));
                if (i2 >= i1) i2++;
            } else {
                i2 = (int)index_of(w[1], (uint32_t)nB);
            }
            p1 = list_get(X, gA, i1);
            p2 = list_get(X, gB, i2);
            v1 = ld_f64(&X.hot[p1].vtot); v2 = ld_f64(&X.hot[p2].vtot);
            m1 = ld_f64(&X.hot[p1].mass); m2 = ld_f64(&X.hot[p2].mass);
            const double k_max = kmax[b1 * B + b2] * fmax;
            const PartProps qa = props_for(ktype, v1, m1, env);
            const PartProps qb = props_for(ktype, v2, m2, env);
            const double k = kernel_pair(ktype, v1, v2, qa, qb, kp);
            const double kf = k * factor;
            over = kf > k_max;
            acc = u53(w[2], w[3]) * k_max < kf; // accept with probability K f / k_max
        }
        const uint64_t m = Grp<L>::ballot(act && acc, sub);
        const int f = m ? __builtin_ctzll(m) : L;
        // status of the candidates actually reached in sequential order
        if (Grp<L>::ballot(act && over && gl <= f, sub)) status |= ST_KMAX_EXCEEDED;
        if (!m) { q0 += L; continue; }

        // ---- coagulate the pair held by lane f (uniform in the group from here) ----
        const int jf = __shfl(j, f, L);
        const uint32_t sf = __shfl(s, f, L);
        const int i1f = __shfl(i1, f, L), i2f = __shfl(i2, f, L);
        const uint32_t p1f = __shfl(p1, f, L), p2f = __shfl(p2, f, L);
        const double v1f = __shfl(v1, f, L), v2f = __shfl(v2, f, L);
        const double m1f = __shfl(m1, f, L), m2f = __shfl(m2, f, L);
        const int b2f = X.L.act[jf];
        const uint32_t nsf = X.L.ns[jf];
        const int gB = c2 * B + b2f;
        const bool same = (gA == gB);

        // slots up to and including this one have started
        tot_samp += sum_ns(counted, jf + 1);

        if (n_ev >= st.max_events) { status |= ST_EVENT_OVERFLOW; done = true; continue; }
        // weighted removal [coagulate_weighting]: with equal weights both quotients are
        // exactly 1 and every uniform in [0,1) passes, so the draw is not needed
        bool rem1 = true, rem2 = true;
        if (n1 != n2) {
            uint32_t w[4];
            rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB, SITE_COAG, sf, w);
            rem1 = u53(w[0], w[1]) < nmin / n1;
            rem2 = u53(w[2], w[3]) < nmin / n2;
        }
        const double vn = v1f + v2f, mn = m1f + m2f;
        const bool big1 = (v1f >= v2f);
        const bool pn_is_1 = (rem1 && rem2) ? big1 : rem1;
        const uint32_t pn = pn_is_1 ? p1f : p2f, po = pn_is_1 ? p2f : p1f;
        const bool rem_o = pn_is_1 ? rem2 : rem1;
        int in = pn_is_1 ? i1f : i2f;
        const int io = pn_is_1 ? i2f : i1f;
        const int g_old = pn_is_1 ? gA : gB, g_o = pn_is_1 ? gB : gA;
        if (vn < edge[0] || vn >= edge[B]) status |= ST_OUT_OF_GRID;
        const int g_new = cc * B + bin_index(edge, B, vn, st);
        if (g_new - cc * B > st.kmax_rng[4 * (int64_t)c + 1]) status |= ST_KMAX_RANGE; // no bound computed there
        if (g_new != g_old && n_app >= st.max_appends) {
            status |= ST_APPEND_OVERFLOW; done = true; continue;
        }
        n_ev++;
        if (st.mlog && lead)
            st.mlog[(int64_t)c * st.max_events + n_ev - 1] = (uint64_t)pn | ((uint64_t)po << 16) | ((uint64_t)pn_is_1 << 32);
        // Source histogram, create times and class of the merged particle
        // [aero_particle_coagulate]: loads go out now and are consumed after the list
        // updates below.  (The species volumes are summed behind the sampler, from the log:
        // k_vol_merge.)
        const int NS = st.n_source;
        double lo_n = 0, lo_o = 0, go_n = 0, go_o = 0;
        if (lead) {
            lo_n = ld_f64(&st.hot[cb + pn].lct); lo_o = ld_f64(&st.hot[cb + po].lct);
            go_n = ld_f64(&st.hot[cb + pn].gct); go_o = ld_f64(&st.hot[cb + po].gct);
        }
        if (rem_o) {
            const int last = list_remove(X, g_o, io, B, lead);
            if (same && in == last) in = io;
            wave_sync();
        }
        if (g_new != g_old) {
            list_remove(X, g_old, in, B, lead);
            wave_sync();
            n_app += list_append(X, g_new, pn, n_app, B, lead);
        }
        if (lead) {
            st_f64(&X.hot[pn].vtot, vn); st_f64(&X.hot[pn].mass, mn);
            if (lo_o < lo_n) st_f64(&st.hot[cb + pn].lct, lo_o);
            if (go_o > go_n) st_f64(&st.hot[cb + pn].gct, go_o);
            if (C > 1) st_u32((uint32_t *)&st.wclass[cb + pn], (uint32_t)cc);
        }
        {
            for (int sp = gl; sp < NS; sp += L) {
                uint32_t *a = (uint32_t *)&norig(st, cb, sp, pn);
                st_u32(a, ld_u32(a) + ld_u32((uint32_t *)&norig(st, cb, sp, po)));
            }
        }
        if (rem_o) {
            if (lead) st_f64(&X.hot[po].vtot, -1.0); // tombstone, in the record just read
            n_dead++;
        }
        if (n_lg >= 0) { // (group-uniform; the leading lane writes)
            const int need = (g_new != g_old ? 1 : 0) + (rem_o ? 1 : 0);
            if (n_lg + need <= EVLOG_CAP) {
                uint32_t *lg = st.evlog + (int64_t)c * EVLOG_CAP;
                if (g_new != g_old) { if (lead) st_u32(&lg[n_lg], pn | ((uint32_t)g_new << 16)); n_lg++; }
                if (rem_o) { if (lead) st_u32(&lg[n_lg], po | 0xFFFF0000u); n_lg++; }
            } else {
                n_lg = -2;
            }
        }
        // (the stores above are waited for just before the next gathers, see "evaluate")
        wave_sync();
        tot_coag++;

        // continue the row after this sample: the window restarts at this slot,
        // later slots are redrawn from the new counts
        ns0 = nsf;
        cur_s0 = sf + 1;
        {
            const int nA = X.L.cnt[gA], nB = X.L.cnt[gB];
            if (nA < 1 || nB < 1 || (same && nA < 2)) cur_s0 = ns0; // exit this slot
        }
        wfrom = b2f + 1;
        keep_first = true;
        need_win = true;
    }

    // The lists are left for the next call (aero_state%valid_sort): lengths go back to HBM;
    // a cell with entries in the overflow list is sorted again next time.
    if (live) {
        wave_sync();
        bool ovf = false;
        for (int g = gl; g < G; g += L) {
            gcnt_g[g] = (int32_t)X.L.cnt[g];
            ovf |= X.L.last_app[g] != 0xFFFF;
        }
        if (ovf) st.sort_valid[c] = 0;
    }
    if (live && lead) {
        if (n_lg >= 0) st.n_log[c] = n_lg;
        else if (n_lg == -2) st.key_valid[c] = 0;
        st.n_events[c] = n_ev;
        st.n_part[c] = st.n_part[c] - n_dead;
        st.n_samp[c] = tot_samp > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)tot_samp;
        st.n_coag[c] = tot_coag;
        if (status) atomicOr(&st.status[c], status);
    }
}

// ------------------------------------------------ coag sampler, lane per cell
//
// The same sweep with one LANE per cell (64 cells per wavefront): every lane runs the
// reference's sequential loop nest for its own cell, so nothing is evaluated
// speculatively and no count is drawn twice.  Every pass of the wavefront's loop a lane
// either evaluates the sample it holds (and coagulates the pair when accepted) or, holding
// none, moves to its next non-empty slots and draws the Poisson counts of up to two of them;
// the memory requests of one kind of lane are issued before the arithmetic of the other
// kind (see the loop).  Everything an accepted pair will need (the list ends that fill the
// holes, first species volume, create times, source count) is requested together with the
// pair's cached volume and mass, so the coagulation itself waits for nothing, and it writes
// only to lines it has read: requests, and above all stores to untouched lines, are what
// bounds this kernel.  A cell's list lengths and offsets live in LDS as one packed
// 32-bit word per group for a window of SEQ_W bins above the cell's first non-empty
// bin (groups beyond it, and entries appended after a group's room ran out, are rare
// and served from HBM); rows of 64 lanes side by side, so the 64 cells never meet on
// an LDS bank.  Chosen for large cell counts, where cells -- not lanes inside a cell --
// are the parallelism (launch_coag_sample).

constexpr int SEQ_W = 24;      // bins per class with list state in LDS
constexpr int SEQ_ENV = 7;     // air properties kept per cell

constexpr size_t SEQ_SHARED_BYTES = 64 * (5 * sizeof(uint32_t) + sizeof(double)); // SPLIT: a lane's sweep state

__host__ __device__ inline size_t seq_lds_bytes(int B, int C, int W = SEQ_W)
{
    size_t o = ((size_t)(B + 1) * sizeof(double) + 15) & ~(size_t)15; // sort-grid edges
    o += sizeof(double) * (POISSON_RCP_N + 2);                        // 1/n of the Poisson walk
    o += sizeof(double) * (SEQ_ENV + (size_t)C) * 64;                 // air properties, weights
    o += sizeof(uint64_t) * 2 * (size_t)C * 64;                       // non-empty bins
    o += sizeof(uint32_t) * (size_t)C * W * 64;                       // count | offset << 16
    return (o + 15) & ~(size_t)15;
}

// diagnostic build only (STAMPS = true, pmcx_set_sampler_variant): s_memtime ticks per phase, summed
// over wavefronts: [0] setup, [1]..[7] the stages of a pass (slots settled and blocks fetched |
// exact counts | pair picked and list entries requested | particle records requested | kernel
// and accept | unused | coagulation), [8] write-back, [9] draws, [10] passes, [11] wavefronts
__device__ unsigned long long g_seq_stamps[12];

// SIMPLE: one weight class known at compile time (the idealised reference initial condition,
// wrf_pmc_init.F90:1766-1770): the weighted-removal draws and the class bookkeeping fold away,
// which is worth registers in the hot loop.  (No sampler touches the species volumes: a pair is
// evaluated from the cached total volume and mass, an event logs its two slots and k_vol_merge
// sums the species behind the sampler in event order -- what the oracle does, pmco.c "apply the
// event log".)
// KT: the coagulation kernel when known at compile time (COAG_KERNEL_TYPE_BROWN, the one the
// reference selects, wrf_pmc_driver.F90:192), 0 = chosen at run time.

// ---------------------------------------- coag sampler, lane per cell, block-wise draws
//
// k_coag_lane.  Once its memory requests were cut (one 32-byte record per particle) the kernel
// was bound by the instructions a wavefront issues per pass: stamped (the STAMPS instantiation),
// more than half of a pass went into finding slots and drawing their counts -- a Philox block
// and an exponential per slot, and a scan through LDS bit masks in nested divergent loops.
// Halving those brought it back to the rate the chip serves random lines at.
// Hence: a lane caches the Philox block and the four bounds of the partner bins gB & ~3 it is
// sweeping; with one class it scans a register word of the non-empty bins in straight-line
// code and settles up to NTRY slots per pass with a compare each (count surely zero); the
// exact Poisson count is taken only for a slot that fails the compare; the accept test of a
// Brownian pair is decided from a single-precision estimate whenever the fp64 kernel cannot
// decide differently (pmcx_device_math.h); the usual event runs in straight-line code.
// Select chains over scalars and by-reference lambda captures are avoided in the loop: the
// compiler turns them into tables and closures in scratch memory, a round trip per use.
// Same results as the oracle's sequential sweep, bit for bit.

// slots a drawing lane may settle per round, rounds (settle, fetch the next block) per pass: one
// wavefront per 64 cells / its two-wavefront form.  Measured on one box each (ms per launch):
//   3*10^5 cells  2 x 4: 3.97-4.10   1 x 6: 3.92   1 x 8: 3.93   1 x 12: 3.94-4.19   3 x 4: 4.11
//   75 000 cells  2 x 4: 1.39-1.42   1 x 6: 1.33   1 x 8: 1.32   1 x 12: 1.38
//   39 000 cells (two wavefronts)  2 x 4: 0.96-0.99   1 x 6, 2 x 6, 1 x 8, 2 x 8: 0.99-1.02
// (with one round the block's bounds are never waited for in the pass that asks for them)
#ifndef PMCX_NTRY
#define PMCX_NTRY 8
#endif
#ifndef PMCX_NROUND
#define PMCX_NROUND 1
#endif
#ifndef PMCX_NTRY_SPLIT
#define PMCX_NTRY_SPLIT 4
#endif
#ifndef PMCX_NROUND_SPLIT
#define PMCX_NROUND_SPLIT 2
#endif
#ifndef PMCX_LANE_WAVES
#define PMCX_LANE_WAVES 3
#endif
// SPLIT: two wavefronts share the 64 cells of a workgroup -- the first evaluates the samples and
// coagulates (stages 1, 3, 5, 7), the second settles the slots (stages 2, 4) -- for blocks of
// cells too few to fill the chip, where a wavefront alone on its SIMD is bound by the
// instructions of its own pass.  What both need of a lane (position, samples left, bound, the
// window word) lives in LDS between passes; in a pass a lane is worked on by exactly one of the
// two (the evaluator while it holds samples, else the drawer), one barrier per pass: per lane
// the sequence of operations, and every result, is the single wavefront's.
// PRE (one class, Brownian: the benchmark's configuration, any species count): the
// draw is split off the sweep.  Phase 0, every lane busy: the lane walks ALL slots of its cell in
// sweep order, four to a Philox block as the sweep would, and keeps the few whose count could be
// non-zero -- tested with both lists PRE_M entries longer than they are now, which bounds their
// length for as long as no bin has taken more than PRE_M merged particles in this call (a list
// only grows by those) -- as 16-byte entries {uniform, bound, slot} in a per-cell list in global
// memory (three slots in four drop out; about 60 of a benchmark cell's 210 remain, 50 of them
// with samples).  Bins that are empty now but could take a merged particle -- gaps in the
// populated range, the two bins above it -- are walked too, as lists of at most PRE_M.  The sweep then takes its slots from that list,
// two entries a load: no bit scan, no block to fetch, no Philox in the draw; a lane goes from one
// possible slot straight to the next (about 95 passes instead of 150, each a third shorter).
// The exact count of an entry is still taken with the lists as they are when the sweep reaches
// it, so every count is the oracle's.  When the list stops describing the cell -- a bin took more
// than PRE_M, a particle was born in a bin phase 0 did not walk, the list ran full -- the lane walks
// the slots behind the one it stands at again, with the lists as they are then (the wavefront
// waits for it: rare).  Only a population that reaches beyond the 32 bins of the window word is left
// to the kernel without PRE, launched behind this one over the cells listed.
// Measured (MI355X, profiles/r3*): passes per wavefront 152 -> 109; at the full domain, where the
// kernel runs at the rate the chip serves random lines, nothing is gained (the list costs lines
// of its own: 4.2 against 3.6 ms), so launch_coag_sample keeps the one-phase form there; on a
// rank's share at N = 4 and 8, where a wavefront's own chain is the floor, 1.24 against 1.46 ms
// (75 000 cells) and 0.77 against 0.99 (39 000).
constexpr int PRE_CAP = 128;  // entries of a cell's list (the last one ends it)
#ifndef PMCX_PRE_M
#define PMCX_PRE_M 12   // arrivals a bin may take before the list stops describing the cell (<= 31)
#endif
constexpr uint32_t PRE_END = 0xFFFFu, PRE_MORE = 0xFFFEu; // slot ids: end of the sweep / list full, go on from .y

// SW: bins of the cell's window with list state in LDS (SEQ_W; 32 -- the whole window word -- in the
// form with the draw split off, which runs where LDS does not limit the wavefronts a SIMD holds)
template <bool STAMPS, bool SIMPLE, int KT, bool SPLIT = false, bool PRE = false, int SW = PRE ? 32 : SEQ_W,
          bool LISTED = false>
__global__ __launch_bounds__(SPLIT ? 128 : 64, SPLIT ? 2 : PMCX_LANE_WAVES)
void k_coag_lane(DeviceState st, int ktype_rt, double del_t, uint64_t seed, uint32_t step,
                uint16_t *ovf_mem, const int32_t *resume_list, int32_t *resume_count, int32_t *next_list = nullptr)
{
    static_assert(!PRE || (SIMPLE && !SPLIT && KT == KERNEL_BROWN), "PRE: one class, Brownian");
    static_assert(!LISTED || (!SIMPLE && !SPLIT && !PRE && !STAMPS), "LISTED: several classes, one wavefront");
    // the second launch: only the cells the first one listed (nothing listed: nothing to do)
    if (!PRE && resume_list && (int)blockIdx.x * 64 >= *resume_count) return;
    uint64_t t_last = 0, t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (STAMPS) t_last = __builtin_amdgcn_s_memtime();
#define STAMP(k)                                                                                  \
    if (STAMPS) {                                                                                 \
        const uint64_t t_now = __builtin_amdgcn_s_memtime();                                      \
        t_acc[k] += t_now - t_last;                                                               \
        t_last = t_now;                                                                           \
    }
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const bool role_e = !SPLIT || (threadIdx.x >> 6) == 0; // evaluates and coagulates
    const bool role_d = !SPLIT || (threadIdx.x >> 6) == 1; // settles the slots
    const bool resuming = !PRE && resume_list != nullptr;
    int cidx = blockIdx.x * 64 + lane;
    if (resuming) cidx = cidx < *resume_count ? resume_list[cidx] : st.n_cells;
    const bool live = cidx < st.n_cells && in_loop(st, cidx);
    const int c = live ? cidx : st.n_cells - 1;
    if (cidx < st.n_cells && !live && role_e) { st.n_samp[cidx] = 0; st.n_coag[cidx] = 0; st.n_events[cidx] = 0; }
    const int ktype = KT ? KT : ktype_rt;
    const int B = st.n_bin, C = SIMPLE ? 1 : st.n_class, G = B * C;
    const int NS = st.n_source;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);

    double *edge = (double *)smem;
    double *rcp = (double *)(smem + (((size_t)(B + 1) * sizeof(double) + 15) & ~(size_t)15));
    double *envs = rcp + (POISSON_RCP_N + 2);
    // (LISTED: the weights, the masks of non-empty bins and the windows of list state are not kept --
    // tens of classes -- and never addressed: list lengths and offsets come from memory)
    double *magl = envs + SEQ_ENV * 64;                   // [C][64] weight magnitudes
    uint64_t *mask = (uint64_t *)(magl + (size_t)C * 64); // [C*2][64]
    uint32_t *ent = (uint32_t *)(mask + 2 * (size_t)C * 64); // [C*SW][64]
    // SPLIT: what the two wavefronts hand each other
    uint32_t *sh_pos = ent + (size_t)C * SW * 64;  // c1 | c2 << 8 | (b1 + 1) << 16 | (b2 + 1) << 24
    uint32_t *sh_ns = sh_pos + 64, *sh_s = sh_ns + 64, *sh_flag = sh_s + 64, *sh_wmask = sh_flag + 64;
    double *sh_kmax = (double *)(sh_wmask + 64);
#define MASK(j) mask[(j) * 64 + lane]

    for (int j = lane; j <= B; j += 64) edge[j] = st.v_edge[j];
    for (int j = lane; j <= POISSON_RCP_N; j += 64) rcp[j] = g_poisson_rcp[j];

    uint16_t *perm = st.perm + (int64_t)c * st.pcap;
    Hot *hot = st.hot + cb;    // cached total volume and mass, create times: one line per particle
    Cold *cold = st.cold + cb; // id, first source count (touched by events with sources only)
    // (per-lane pointers cost two registers each: the rarely used ones are formed on use)
#define gcnt_g (st.gcnt + (int64_t)c * G)
#define goff_g (st.goff + (int64_t)c * (G + 1))
#define O_ovf_n(g) (ovf_mem[(size_t)c * (2 * (size_t)G + 2 * (size_t)st.max_appends) + (g)])
#define O_last_app(g) (ovf_mem[(size_t)c * (2 * (size_t)G + 2 * (size_t)st.max_appends) + G + (g)])
#define O_app_slot(e) (ovf_mem[(size_t)c * (2 * (size_t)G + 2 * (size_t)st.max_appends) + 2 * G + (e)])
#define O_app_prev(e) (ovf_mem[(size_t)c * (2 * (size_t)G + 2 * (size_t)st.max_appends) + 2 * G + st.max_appends + (e)])

    // non-empty bins of every class; the window starts at the cell's first one
    // (the list lengths are requested eight at a time, the window's entries all at once: one
    // request per wait made the set-up 180 memory round trips, 8 % of the kernel)
    int blo = B;
    if (!LISTED)
    for (int cl = 0; cl < C; cl++) {
        uint64_t m0 = 0, m1 = 0;
        for (int b0 = 0; b0 < B; b0 += 8) {
            int32_t n8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) n8[k] = gcnt_g[cl * B + min(b0 + k, B - 1)];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int b = b0 + k; // (uniform)
                const bool has = b < B && n8[k] > 0;
                const uint64_t bit = has ? 1ull << (b & 63) : 0ull;
                if (b < 64) m0 |= bit; else m1 |= bit;
                blo = (has && b < blo) ? b : blo;
            }
        }
        MASK(cl * 2) = m0; MASK(cl * 2 + 1) = m1;
    }
    if (blo > B - SW) blo = (B - SW > 0) ? B - SW : 0;
    if (!LISTED)
    for (int cl = 0; cl < C; cl++) {
        uint32_t e[SW];
#pragma unroll
        for (int k = 0; k < SW; k++) {
            const int b = min(blo + k, B - 1);
            e[k] = (uint32_t)gcnt_g[cl * B + b] | ((uint32_t)goff_g[cl * B + b] << 16);
        }
#pragma unroll
        for (int k = 0; k < SW; k++)
            ent[(cl * SW + k) * 64 + lane] = (blo + k < B) ? e[k] : 0u;
    }
    if (!LISTED) for (int cl = 0; cl < C; cl++) magl[cl * 64 + lane] = st.mag[(int64_t)c * C + cl];
    {
        const EnvDerived e = env_derive(st.temp[c], st.pressure[c]);
        envs[0 * 64 + lane] = e.gasfreepath; envs[1 * 64 + lane] = e.inv_gfp;
        envs[2 * 64 + lane] = e.c_diff; envs[3 * 64 + lane] = e.c_speed;
        envs[4 * 64 + lane] = e.c_inv_speed; envs[5 * 64 + lane] = e.c_fall;
        envs[6 * 64 + lane] = e.rhoair;
        // The last bin the cell's bounds cover, looked at when an event creates a particle.  The
        // Brownian kernel does not use the air density: its slot carries it (another 512 bytes
        // of LDS per wavefront would cost the twelfth wavefront of the CU).
        if (KT == KERNEL_BROWN) envs[6 * 64 + lane] = (double)st.kmax_rng[4 * (int64_t)c + 1];
    }
    __syncthreads(); // one wavefront: orders the edge table

    const KernelParams kp{st.const_k, st.additive_beta};

    // ---- list state: packed entry = count | offset << 16 ----
    auto in_win = [&](int b) -> bool { return !LISTED && (unsigned)(b - blo) < (unsigned)SW; };
    auto ent_get = [&](int cl, int b) -> uint32_t {
        if (in_win(b)) return ent[(cl * SW + (b - blo)) * 64 + lane];
        const int g = cl * B + b;
        return (uint32_t)gcnt_g[g] | ((uint32_t)goff_g[g] << 16);
    };
    // One class: the non-empty bins blo .. blo + 31 as a register word (bit k: bin blo + k), what
    // the sweep scans (32 bins are 3.2 decades of radius).  `far`: some particle lives beyond
    // even that (the general scan serves); `wide`: some particle lives outside the LDS window
    // (its list length comes from memory).
    uint32_t wmask = 0;
    bool far = false, wide = false;
    // Copies of the lane and cell numbers the compiler cannot see through, refreshed every pass
    // in front of the event code: what an event forms from them (list-state addresses in LDS, the
    // cell's place in the event log) is then computed where it is used.  Formed from `lane` and
    // `c` these are loop invariants, hoisted out of the loop and -- the loop being short of
    // registers -- spilled: six scratch reloads per event, each behind a wait for every
    // store the event had just issued.
    int lane_c = lane, c_c = c;
#define MASKC(j) mask[(j) * 64 + lane_c]
    auto cnt_set = [&](int cl, int b, uint32_t e, int n) { // new length n of the group with entry e
        if (in_win(b)) {
            ent[(cl * SW + (b - blo)) * 64 + lane_c] = (e & 0xFFFF0000u) | (uint32_t)n;
        } else {
            gcnt_g[cl * B + b] = n;
            if (n > 0) wide = true;
        }
        if (SIMPLE) {
            if ((unsigned)(b - blo) < 32u) {
                if (n == 0) wmask &= ~(1u << (b - blo));
                else wmask |= 1u << (b - blo);
            } else if (n > 0) {
                far = true;
            }
        }
        if (LISTED) return;
        if (n == 0) MASKC(cl * 2 + (b >> 6)) &= ~(1ull << (b & 63));
        else if (n == 1) MASKC(cl * 2 + (b >> 6)) |= 1ull << (b & 63);
    };
    auto room_end = [&](int cl, int b) -> int { // offset where the room behind the group ends
        if (b + 1 < B && in_win(b + 1)) return (int)(ent[(cl * SW + (b + 1 - blo)) * 64 + lane] >> 16);
        return goff_g[cl * B + b + 1];
    };
    // entries beyond a group's room (rare): linked through the per-cell table
    auto ovf_of = [&](int g, int n_app) -> int { return n_app ? (int)O_ovf_n(g) : 0; };
    auto list_get = [&](int g, uint32_t e, int i, int n_app) -> uint32_t {
        const int n = (int)(e & 0xFFFFu), nb = n - ovf_of(g, n_app);
        if (i < nb) return perm[(e >> 16) + i];
        int x = O_last_app(g);
        for (int k = n - 1 - i; k > 0; k--) x = O_app_prev(x);
        return O_app_slot(x);
    };
    auto list_set = [&](int g, uint32_t e, int i, uint32_t val, int n_app) {
        const int n = (int)(e & 0xFFFFu), nb = n - ovf_of(g, n_app);
        if (i < nb) {
            perm[(e >> 16) + i] = val;
        } else {
            int x = O_last_app(g);
            for (int k = n - 1 - i; k > 0; k--) x = O_app_prev(x);
            O_app_slot(x) = (uint16_t)val;
        }
    };
    auto next_bin = [&](int cl, int from) -> int { // next non-empty bin >= from, or -1
        if (from < 64) {
            const uint64_t w = MASK(cl * 2) & (~0ull << from);
            if (w) return __builtin_ctzll(w);
            from = 64;
        }
        if (from < 128) {
            const uint64_t w = MASK(cl * 2 + 1) & (~0ull << (from - 64));
            if (w) return 64 + __builtin_ctzll(w);
        }
        return -1;
    };

    int n_ev = 0, n_app = 0, n_dead = 0, tot_coag = 0, status = 0;
    uint64_t tot_samp = 0;
    // the cell's event log (what the next sort needs to bring the slots' keys up to date):
    // entries so far, -1: the keys do not hold anyway, -2: the log ran over (they no longer do)
    int n_lg = st.key_valid[c] != 0 ? st.n_log[c] : -1;
#define PMCX_LOG(slot, key)                                                                         \
    if (n_lg >= 0) {                                                                              \
        if (n_lg < EVLOG_CAP) st.evlog[(int64_t)c_c * EVLOG_CAP + n_lg++] = (uint32_t)(slot) | ((uint32_t)(key) << 16); \
        else n_lg = -2;                                                                           \
    }
    int c1 = 0, c2 = 0, b1 = -1, b2 = -1, cc = 0;
    double n1 = 0, n2 = 0, nmin = 0, fmax = 0, factor = 0, k_max = 0;
    uint32_t ns_rem = 0, s = 0;
    bool done = !live;
    if (resuming && live) {
        // where the first launch left the cell (a slot boundary): what it counted so far, the slot
        // it settled last, the entries it had put in the table past the groups' rooms
        const int32_t pos = st.resume[4 * (int64_t)c];
        n_app = st.resume[4 * (int64_t)c + 1];
        n_ev = st.n_events[c]; tot_coag = st.n_coag[c]; tot_samp = (uint64_t)(uint32_t)st.n_samp[c];
        if (pos != -1) { b1 = pos & 0xFF; b2 = (pos >> 8) & 0xFF; c1 = (pos >> 16) & 0xFF; c2 = (pos >> 24) & 0xFF; }
    }
    if (SIMPLE) {
        // bins blo .. blo + 63 of the 128-bit mask (m1 : m0), shifted down by blo
        const uint64_t m0 = MASK(0), m1 = MASK(1);
        uint64_t w64;
        if (blo == 0) w64 = m0;
        else if (blo < 64) w64 = (m0 >> blo) | (m1 << (64 - blo));
        else w64 = m1 >> (blo - 64);
        wmask = (uint32_t)w64;
        // (bins below blo hold nothing: blo is the first non-empty bin unless the grid is
        // narrower than the window; bins from blo + 32 on are beyond the word)
        far = (w64 >> 32) != 0 || (blo < 64 && (m1 >> blo) != 0);
        if (blo > 0 && blo < 64 && (m0 & ((1ull << blo) - 1ull))) far = true;
        if (blo >= 64 && (m0 || (m1 & ((1ull << (blo - 64)) - 1ull)))) far = true;
        wide = far || (SW < 32 && ((uint64_t)wmask >> SW) != 0);
    }
    STAMP(0)

    // ---- where the lane stands in the sweep, what it holds, the Philox block and the bounds of
    // ---- the four partner bins it is drawing from ----
    constexpr uint32_t TAG_NONE = 0xFFFFFFFFu;
    constexpr int NTRY = SPLIT ? PMCX_NTRY_SPLIT : PMCX_NTRY;  // slots a drawing lane may settle per round
    // rounds (settle, fetch the next block) per pass; in the two-wavefront form, measured at 37 500
    // cells: 1 round 1.05 ms, 2 rounds 0.96, 3 rounds 1.07, 4 rounds 1.21 (the drawer is on the
    // pass's critical path as much as the evaluator)
    constexpr int NROUND = SPLIT ? PMCX_NROUND_SPLIT : PMCX_NROUND;
    uint32_t blk_tag = TAG_NONE;     // (gA << 16) | (gB & ~3) of the cached block
    // its four words (the uniforms of slots gB & 3) and kmax[b1][b2] of the same four partner
    // bins, two to a 64-bit value: picked with shifts (a chain of selects over four scalars is
    // turned into a table in scratch memory by the compiler, a memory round trip per slot)
    uint64_t wq01 = 0, wq23 = 0, kq01 = 0, kq23 = 0;
    int rc1 = -1, rc2 = -1;          // classes the row constants below belong to
    float cf_up = 0.0f;              // one class: an upper bound of fmax * del_t in single precision
    if (SIMPLE) { // one class: the row constants are the cell's
        n1 = n2 = nmin = fmax = factor = magl[lane];
        cc = 0; rc1 = rc2 = 0;
        cf_up = (float)(fmax * del_t) * 1.000001f;
        if (resuming && live && b1 >= 0) {
            // left in the middle of a slot (k_coag_grp hands a cell over in front of an event its
            // straight-line code does not cover): the samples still to evaluate, the next one's index
            ns_rem = (uint32_t)st.resume[4 * (int64_t)c + 2];
            s = (uint32_t)st.resume[4 * (int64_t)c + 3];
            if (ns_rem) k_max = (double)st.kmax[(int64_t)c * st.kmax_stride + b1 * B + b2] * fmax;
        }
    }

    // The slot the lane would draw next from where it stands, with the lists as they are now:
    // the sweep order c1, c2 <= c1, b1, b2 (>= b1 within a class), skipping what cannot give a
    // pair.  Does not move the lane (the caller commits).  False: the sweep is over.
    auto find_next = [&](int &t1, int &t2, int &tb1, int &tb2, uint32_t &eA, uint32_t &eB)
        __attribute__((always_inline)) -> bool {
        t1 = c1; t2 = c2; tb1 = b1; tb2 = b2;
        while (true) {
            if (tb1 >= 0) {
                eA = ent_get(t1, tb1);
                if ((eA & 0xFFFFu) > 0) {
                    const int nb2 = next_bin(t2, tb2 + 1);
                    if (nb2 >= 0 && nb2 < B) {
                        tb2 = nb2;
                        const bool same = (t1 == t2 && tb1 == tb2);
                        eB = same ? eA : ent_get(t2, tb2);
                        if (!(same && (eA & 0xFFFFu) < 2)) return true;
                        continue;
                    }
                }
            }
            tb1 = next_bin(t1, tb1 + 1);
            if (tb1 < 0 || tb1 >= B) {
                tb1 = -1;
                if (++t2 > t1) { t2 = 0; ++t1; }
                if (t1 >= C) return false;
                continue;
            }
            tb2 = ((t1 == t2) ? tb1 : 0) - 1;
        }
    };

    // Every pass of the loop a lane either evaluates one sample (and coagulates the pair when
    // accepted) or, holding none, settles up to NTRY slots.  A slot's Poisson count is zero
    // whenever its uniform lies below 1 - mean (<= exp(-mean)): that test needs neither the
    // exponential nor a new Philox block while the lane stays within the four partner bins of
    // its cached block, so most slots cost a bit scan, two LDS words and a compare; a slot that
    // fails it takes the exact count, and a lane that leaves its block fetches the next one (one
    // Philox block, four bounds) for the following pass.  Memory requests of one kind of lane
    // are issued before the arithmetic of the other kind.
    // ---- PRE: the cell's list of the slots whose count could be non-zero (made in the loop below,
    // ---- at the start and again whenever it stops describing the cell) ----
    uint4 *pre = nullptr;            // the cell's list
    uint32_t pmask0 = 0;             // the bins walked when it was made
    uint4 ea = make_uint4(0, 0, PRE_END, 0), eb = ea; // the two entries in hand (.x uniform, .y bound, .z slot)
    int pk = 0, pk_pair = 0;         // entries taken so far; the pair in hand starts there
    bool pre_stop = false;           // the list no longer describes the cell
    bool unfinished = false;
    bool need_pre = false;           // the lane is to walk its slots (again) from where it stands
    uint64_t arr0 = 0, arr1 = 0;     // merged particles each bin of the window has taken (5 bits a bin)
    // ---- LISTED (several classes): the slots whose count could be non-zero come from the cell's list,
    // ---- made by k_coag_screen for as long as no group outgrows the length it was tested with
    const uint2 *sl = nullptr;
    uint2 lq[4];                         // the four entries in hand
    int lk = 0;
    if (LISTED) {
        sl = st.slist + (int64_t)c * SLIST_CAP;
#pragma unroll
        for (int k = 0; k < 4; k++) lq[k] = done ? make_uint2(SL_END, 0u) : sl[k];
        wide = true;                     // (list lengths and offsets from memory)
    }
    if (PRE) {
        pre = st.pre + (int64_t)c * PRE_CAP;
        if (live && far) { unfinished = true; done = true; }  // (beyond the 32 bins of the window word: the other kernel)
        if (done) { far = false; wide = false; }   // (these pick the wavefront's paths: finished lanes do not vote)
        need_pre = !done;
    }

    if (SPLIT) { // (both wavefronts write the same values)
        sh_pos[lane] = 0; sh_ns[lane] = 0; sh_s[lane] = 0; sh_kmax[lane] = 0.0;
        sh_flag[lane] = (done ? 1u : 0u) | (far ? 2u : 0u) | (wide ? 4u : 0u);
        sh_wmask[lane] = wmask;
        __syncthreads();
    }
    while (true) {
        bool own = true; // SPLIT: this wavefront works on the lane in this pass
        if (SPLIT) {
            const uint32_t pk = sh_pos[lane], fl = sh_flag[lane];
            c1 = (int)(pk & 255u); c2 = (int)((pk >> 8) & 255u);
            b1 = (int)((pk >> 16) & 255u) - 1; b2 = (int)(pk >> 24) - 1;
            ns_rem = sh_ns[lane]; s = sh_s[lane]; k_max = sh_kmax[lane];
            done = (fl & 1u) != 0; far = (fl & 2u) != 0; wide = (fl & 4u) != 0;
            wmask = sh_wmask[lane];
            if (!SIMPLE) { // what the scan derives from the classes of the slot
                n1 = magl[(c1 < C ? c1 : C - 1) * 64 + lane]; n2 = magl[(c2 < C ? c2 : C - 1) * 64 + lane];
                cc = (n1 < n2) ? c1 : c2;
                nmin = (n1 < n2) ? n1 : n2;
                fmax = (n1 < n2) ? n2 : n1;
                factor = n1 * n2 / nmin;
                rc1 = c1; rc2 = c2;
            }
            own = !done && (role_e ? ns_rem > 0 : ns_rem == 0);
        }
        if (__ballot(!done) == 0) break;
        if (STAMPS) t_acc[10]++;
        // ---- PRE, phase 0: the lane walks the slots behind (b1, b2) in sweep order, four to a Philox
        // ---- block, and lists those whose count could be non-zero while no bin takes more than
        // ---- PRE_M arrivals.  All lanes at the start; later a lane whose list stopped holding.
        if (PRE && __ballot(need_pre) != 0) {
            if (need_pre) {
                // the bins walked: the populated ones, the gaps between them, two above the last
                // (the window's last bin cannot take an arrival on the fast path)
                {
                    const int top = 31 - __builtin_clz(wmask | 1u), lo = __builtin_ctz(wmask | 0x80000000u);
                    int hi = top + 2;
                    if (hi > SW - 2) hi = SW - 2;
                    if (hi < top) hi = top;
                    pmask0 = wmask | ((hi >= 31 ? ~0u : ((1u << (hi + 1)) - 1u)) & (~0u << lo));
                    if (blo + 32 > B) pmask0 &= (B - blo >= 32) ? ~0u : ((1u << (B - blo)) - 1u);
                }
                const float *kmx = st.kmax + (int64_t)c * st.kmax_stride;
                int npre = 0;
                int32_t prev = b1 < 0 ? -1 : (b1 | (b2 << 8));   // the slot tested last
                bool full = false;
                const int r1s = b1 < 0 ? -1 : b1 - blo, r2s = b2 - blo;   // (walk what lies behind this slot)
                uint32_t rows = r1s < 0 ? pmask0 : (pmask0 & (~0u << r1s));
                while (rows && !full) {
                    const int r1 = __builtin_ctz(rows);
                    rows &= rows - 1;
                    const int pb1 = blo + r1;
                    const double nA_up = (double)((ent_get(0, pb1) & 0xFFFFu) + (uint32_t)PMCX_PRE_M);
                    uint32_t pm = pmask0 & (~0u << r1);
                    if (r1 == r1s) pm = r2s >= 31 ? 0u : (pm & (~0u << (r2s + 1)));
                    while (pm && !full) {
                        const int q4 = (blo + __builtin_ctz(pm)) & ~3;        // the block's first bin
                        uint32_t w[4];
                        rng_block(seed, cell_id, step, ((uint32_t)pb1 << 16) | (uint32_t)q4, SITE_POISSON, 0, w);
                        const float4 kf4 = *(const float4 *)(kmx + pb1 * B + q4);
                        const float kfa[4] = {kf4.x, kf4.y, kf4.z, kf4.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int pb2 = q4 + k, r2 = pb2 - blo;
                            if (r2 >= r1 && r2 < 32 && ((pm >> r2) & 1u) && !full) {
                                const double nB_up = (double)((ent_get(0, pb2) & 0xFFFFu) + (uint32_t)PMCX_PRE_M);
                                const double np_up = (pb2 == pb1) ? nA_up * (nA_up - 1.0) / 2.0 : nA_up * nB_up;
                                const double mean_up = ((double)kfa[k] * fmax) * del_t * np_up; // (the sweep's expression)
                                const double u_ = (double)w[k] * 0x1.0p-32;
                                if (!(u_ < (1.0 - mean_up) - 4e-15)) {
                                    if (npre < PRE_CAP - 1) {
                                        pre[npre++] = make_uint4(w[k], __float_as_uint(kfa[k]),
                                                                 (uint32_t)pb1 | ((uint32_t)pb2 << 8), 0u);
                                    } else {
                                        full = true;   // (go on from the slot tested before this one)
                                    }
                                }
                                if (!full) prev = pb1 | (pb2 << 8);
                            }
                        }
                        const int end = q4 + 4 - blo;                          // bins below it are done
                        pm = end >= 32 ? 0u : (pm & (~0u << end));
                    }
                }
                pre[npre] = full ? make_uint4(0u, (uint32_t)prev, PRE_MORE, 0u) : make_uint4(0u, 0u, PRE_END, 0u);
            }
            // (a lane reads what it wrote itself: its stores are ordered before its loads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (need_pre) {
                ea = pre[0]; eb = pre[1];
                pk = 0; pk_pair = 0; arr0 = 0; arr1 = 0;
                pre_stop = false; need_pre = false;
            }
            STAMP(6)   // (phase 0 in the stage-6 slot of the diagnostic build)
        }
        // (uniform: they pick the scan and the list-length reads for the whole wavefront)
        const bool any_far = !SIMPLE || __ballot(far) != 0;
        const bool any_wide = any_far || __ballot(wide) != 0;

        // The stages of a pass: a lane holding no samples settles slots (DRAW) and takes the exact
        // count of the one that needs it (COUNT); a lane holding samples -- also the ones just
        // drawn -- evaluates one (PICK: the pair, its list entries requested; RECORDS: the two
        // particle records requested; kernel and accept; coagulate).  Drawing first lets a lane
        // evaluate the first sample of a slot in the pass that drew it: 140 passes for the
        // benchmark's cells instead of 200 (measured on one box, 3*10^5 cells: 4.19-4.24 ms
        // against 4.50 with PICK in front of DRAW; 1.41 against 1.70 at 75 000 cells).
        bool evalid = false, acc = false;
        int ev_bnew = -1;                       // PRE: the bin an event's merged particle moved into
        int i1 = 0, i2 = 0;
        uint32_t p1 = 0, p2 = 0, s_ev = 0, eA = 0, eB = 0;
        uint32_t last1 = 0, last2 = 0;          // entries at the ends of the two lists
        uint32_t last1b = 0;                    // a pair within one list: the entry in front of its end
        double u_acc = 0;
        int gA = 0, gB = 0;
        bool same = false;
        double v1 = 0, v2 = 0, m1 = 0, m2 = 0;
        double lct1 = 0, lct2 = 0, gct1 = 0, gct2 = 0; // create times
        int32_t no1 = 0, no2 = 0;               // first source

        // ---- drawing lanes (no samples in hand): settle slots inside the cached block, fetch the
        // ---- next block when the sweep leaves it, settle on -- NROUND rounds a pass ----
        bool pending = false;            // a slot whose count needs the exact Poisson
        double p_mean = 0, p_u = 0;
        uint32_t pe_A = 0, pe_B = 0;     // LISTED: the drawn slot's list entries (count | offset << 16)
        bool have_ent = false;
        bool dr = role_d && own && !done && ns_rem == 0;
        if (STAMPS && dr) t_acc[9]++;
        // the slot's mean and uniform from the cached block; ZERO = true: its count is surely zero
#define PMCX_SETTLE(hA, hB, dA, dB, ZERO)                                                           \
    {                                                                                             \
        const uint32_t sh_ = ((hB) & 1u) * 32u;                                                   \
        const uint32_t wu_ = (uint32_t)((((hB) & 2u) ? wq23 : wq01) >> sh_);                      \
        const float kf_ = __uint_as_float((uint32_t)((((hB) & 2u) ? kq23 : kq01) >> sh_));       \
        const double nA_ = (double)((dA) & 0xFFFFu), nB_ = (double)((dB) & 0xFFFFu);              \
        k_max = (double)kf_ * fmax;                                                               \
        const double n_possible_ = ((hA) == (hB)) ? nA_ * (nA_ - 1.0) / 2.0 : nA_ * nB_;          \
        const double mean_ = k_max * del_t * n_possible_;                                         \
        const double u_ = (double)wu_ * 0x1.0p-32;                                                \
        /* u < 1 - mean - 4e-15  =>  u < det_exp(-mean): no sample, exactly as poisson() says */  \
        ZERO = u_ < (1.0 - mean_) - 4e-15;                                                        \
        if (!ZERO) { pending = true; p_mean = mean_; p_u = u_; }                                  \
    }
        if (PRE) {
            // the next slots from the cell's list: two entries in hand, two tries a pass
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (__ballot(dr) == 0) break;
                if (pk >= pk_pair + 2) dr = false;   // (both entries in hand are used: the next pair is on its way)
                if (dr) {
                    if (pre_stop) {            // the list no longer describes the cell
                        if (far) { unfinished = true; done = true; far = false; wide = false; }  // (the other kernel goes on)
                        else need_pre = true;  // it is made again from here, at the top of the next pass
                    } else {
                        const uint4 e = (pk & 1) ? eb : ea;
                        if (e.z == PRE_END) {
                            done = true;
                        } else if (e.z == PRE_MORE) {  // the list ran full: made again from the slot it reached
                            const int32_t pv = (int32_t)e.y;
                            if (pv >= 0) { b1 = pv & 0xFF; b2 = (pv >> 8) & 0xFF; } else { b1 = -1; b2 = -1; }
                            need_pre = true;
                        } else {
                            b1 = (int)(e.z & 0xFFu); b2 = (int)((e.z >> 8) & 0xFFu);
                            uint32_t dA, dB;
                            if (!any_wide) { dA = ent[(b1 - blo) * 64 + lane]; dB = ent[(b2 - blo) * 64 + lane]; }
                            else { dA = ent_get(0, b1); dB = ent_get(0, b2); }
                            const double nA_ = (double)(dA & 0xFFFFu), nB_ = (double)(dB & 0xFFFFu);
                            k_max = (double)__uint_as_float(e.y) * fmax;
                            const double n_possible_ = (b1 == b2) ? nA_ * (nA_ - 1.0) / 2.0 : nA_ * nB_;
                            const double mean_ = k_max * del_t * n_possible_;
                            const double u_ = (double)e.x * 0x1.0p-32;
                            // (a list that ran out gives n_possible <= 0: no sample, as the sweep has it)
                            if (!(u_ < (1.0 - mean_) - 4e-15)) { pending = true; p_mean = mean_; p_u = u_; }
                            pk++;
                        }
                    }
                }
                dr = dr && !done && !pending && !need_pre;
            }
            // both entries used: the next two (they arrive while the wavefront evaluates)
            if (!done && !need_pre && pk >= pk_pair + 2) {
                ea = pre[pk]; eb = pre[pk + 1];
                pk_pair = pk;
            }
        } else if (LISTED) {
            // The next slots from the cell's list, up to four a pass: the list lengths, the bounds and the
            // weights of all four are requested together, then each entry's exact count is taken in order
            // with the lists as they are now, up to the first that holds samples (nothing changes the
            // lists in between: entries that draw none cost a quarter of a pass each).
            if (dr) {
                if (pre_stop) {
                    // a group outgrew the length the list was made for (the slots it left out may draw
                    // samples now): the next round lists what lies behind the slot settled last
                    unfinished = true; done = true;
                } else {
                    uint32_t cA[4], cB[4], oA[4], oB[4];
                    float kf[4];
                    double wa[4], wb[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t x = lq[k].x < SL_FULL ? lq[k].x : 0u; // c1 << 24 | b1 << 16 | c2 << 8 | b2
                        const int e1 = (int)(x >> 24), d1 = (int)((x >> 16) & 0xFFu), e2 = (int)((x >> 8) & 0xFFu), d2 = (int)(x & 0xFFu);
                        cA[k] = (uint32_t)gcnt_g[e1 * B + d1];
                        cB[k] = (uint32_t)gcnt_g[e2 * B + d2];
                        oA[k] = (uint32_t)goff_g[e1 * B + d1];   // (for the first sample of the entry that holds some)
                        oB[k] = (uint32_t)goff_g[e2 * B + d2];
                        kf[k] = st.kmax[(int64_t)c * st.kmax_stride + d1 * B + d2];
                        wa[k] = st.mag[(int64_t)c * C + e1];
                        wb[k] = st.mag[(int64_t)c * C + e2];
                    }
                    int used = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (!pending && !done) {
                            const uint32_t x = lq[k].x;
                            if (x == SL_END) {
                                done = true;
                            } else if (x == SL_FULL) {   // the list ran full: the next round goes on from here
                                unfinished = true; done = true;
                            } else {
                                c1 = (int)(x >> 24); b1 = (int)((x >> 16) & 0xFFu); c2 = (int)((x >> 8) & 0xFFu); b2 = (int)(x & 0xFFu);
                                n1 = wa[k]; n2 = wb[k];
                                cc = (n1 < n2) ? c1 : c2;
                                nmin = (n1 < n2) ? n1 : n2;
                                fmax = (n1 < n2) ? n2 : n1;
                                const double nA_ = (double)(cA[k] & 0xFFFFu), nB_ = (double)(cB[k] & 0xFFFFu);
                                k_max = (double)kf[k] * fmax;
                                const double n_possible_ = (c1 == c2 && b1 == b2) ? nA_ * (nA_ - 1.0) / 2.0 : nA_ * nB_;
                                const double mean_ = k_max * del_t * n_possible_;
                                const double u_ = (double)lq[k].y * 0x1.0p-32;
                                // (a list that ran out gives n_possible <= 0: no sample, as the sweep has it)
                                if (!(u_ < (1.0 - mean_) - 4e-15)) {
                                    pending = true; p_mean = mean_; p_u = u_;
                                    pe_A = (cA[k] & 0xFFFFu) | (oA[k] << 16); pe_B = (cB[k] & 0xFFFFu) | (oB[k] << 16);
                                    have_ent = true;
                                }
                                used++;
                            }
                        }
                    }
                    if (pending) factor = n1 * n2 / nmin;
                    lk += used;
                    if (!done) {
#pragma unroll
                        for (int k = 0; k < 4; k++) lq[k] = sl[min(lk + k, SLIST_CAP - 1)];
                    }
                }
            }
        } else
        for (int r = 0; r < NROUND; r++) {
            if (__ballot(dr) == 0) break;
            uint32_t want = TAG_NONE;    // the block of the next slot, when it is not the cached one
            if (SIMPLE && !any_far) {
                // One class, every particle inside the window: straight-line code, NTRY times.  The
                // next slot of the sweep is the next set bit of the window word behind b2, or the
                // first bin of the next populated row with itself as partner; a bin that holds one
                // particle gives n_possible = 0 as its own partner, i.e. a mean of zero.
                bool active = dr;
#pragma unroll
                for (int t = 0; t < NTRY; t++) {
                    if (active) {
                        // (shifts by 32 are avoided: a position at bit 31 has nothing behind it)
                        const int p1_ = b1 - blo, p2_ = b2 - blo;
                        const bool rowok = b1 >= 0 && ((wmask >> p1_) & 1u);
                        const uint32_t m = (rowok && p2_ < 31) ? (wmask & (~0u << (p2_ + 1))) : 0u;
                        const uint32_t mr = b1 < 0 ? wmask : (p1_ < 31 ? (wmask & (~0u << (p1_ + 1))) : 0u);
                        const int nb1 = m ? b1 : blo + __builtin_ctz(mr | 0x80000000u);
                        const int nb2 = m ? blo + __builtin_ctz(m) : nb1;
                        if (!m && !mr) { done = true; active = false; }
                        else {
                            const uint32_t tag = ((uint32_t)nb1 << 16) | ((uint32_t)nb2 & ~3u);
                            if (tag != blk_tag) { want = tag; active = false; }
                            else {
                                b1 = nb1; b2 = nb2;
                                uint32_t dA, dB;
                                if (!any_wide) {
                                    dA = ent[(nb1 - blo) * 64 + lane];
                                    dB = ent[(nb2 - blo) * 64 + lane];
                                } else {
                                    dA = ent_get(0, nb1);
                                    dB = ent_get(0, nb2);
                                }
                                // Surely no sample: decided in single precision from upper bounds of
                                // the uniform and of the mean (u <= u_up < 1 - mean_up - 1e-6 implies
                                // the double-precision test of PMCX_SETTLE, hence a count of zero);
                                // a slot that fails it -- the one with samples, or one within 1e-6 of
                                // the line -- takes the double-precision values and the exact count.
                                bool zero;
                                {
                                    const uint32_t hB_ = (uint32_t)nb2, sh_ = (hB_ & 1u) * 32u;
                                    const uint32_t wu_ = (uint32_t)(((hB_ & 2u) ? wq23 : wq01) >> sh_);
                                    const float kf_ = __uint_as_float((uint32_t)(((hB_ & 2u) ? kq23 : kq01) >> sh_));
                                    const uint32_t nA_ = dA & 0xFFFFu, nB_ = dB & 0xFFFFu;
                                    const uint32_t np_ = nb1 == nb2 ? (nA_ * (nA_ - 1u)) >> 1 : nA_ * nB_;
                                    const float mean_up = (kf_ * cf_up) * ((float)np_ * 1.00001f);
                                    const float u_up = (float)((wu_ >> 8) + 1u) * 0x1.0p-24f;
                                    zero = u_up < (1.0f - mean_up) - 1e-6f;
                                }
                                if (!zero) {
                                    PMCX_SETTLE((uint32_t)nb1, (uint32_t)nb2, dA, dB, zero)
                                    if (!zero) active = false;
                                }
                            }
                        }
                    }
                }
            } else if (dr) {
                // the general scan (several classes, or a particle outside the window): one slot a round
                int t1, t2, tb1, tb2;
                uint32_t dA = 0, dB = 0;
                if (!find_next(t1, t2, tb1, tb2, dA, dB)) {
                    done = true;
                } else {
                    const uint32_t hA = (uint32_t)(t1 * B + tb1), hB = (uint32_t)(t2 * B + tb2);
                    const uint32_t tag = (hA << 16) | (hB & ~3u);
                    if (tag != blk_tag) {
                        want = tag;
                    } else {
                        c1 = t1; c2 = t2; b1 = tb1; b2 = tb2;
                        if (!SIMPLE && (c1 != rc1 || c2 != rc2)) {
                            n1 = magl[c1 * 64 + lane]; n2 = magl[c2 * 64 + lane];
                            cc = (n1 < n2) ? c1 : c2;
                            nmin = (n1 < n2) ? n1 : n2;
                            fmax = (n1 < n2) ? n2 : n1;
                            factor = n1 * n2 / nmin;
                            rc1 = c1; rc2 = c2;
                        }
                        bool zero;
                        PMCX_SETTLE(hA, hB, dA, dB, zero)
                        (void)zero;
                    }
                }
            }
            // the block the sweep moves into: one Philox block, four bounds
            if (want != TAG_NONE) {
                uint32_t w[4];
                rng_block(seed, cell_id, step, want, SITE_POISSON, 0, w);
                wq01 = (uint64_t)w[0] | ((uint64_t)w[1] << 32); wq23 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
                const int hA = (int)(want >> 16), hBq = (int)(want & 0xFFFFu);
                const int rb1 = SIMPLE ? hA : hA % B;
                const float *row = st.kmax + (int64_t)c * st.kmax_stride + rb1 * B;
                // (a group index beyond the last one belongs to no slot: its bound is never read)
                const float kq0 = row[SIMPLE ? min(hBq, B - 1) : (hBq % B)];
                const float kq1 = row[SIMPLE ? min(hBq + 1, B - 1) : ((hBq + 1) % B)];
                const float kq2 = row[SIMPLE ? min(hBq + 2, B - 1) : ((hBq + 2) % B)];
                const float kq3 = row[SIMPLE ? min(hBq + 3, B - 1) : ((hBq + 3) % B)];
                kq01 = (uint64_t)__float_as_uint(kq0) | ((uint64_t)__float_as_uint(kq1) << 32);
                kq23 = (uint64_t)__float_as_uint(kq2) | ((uint64_t)__float_as_uint(kq3) << 32);
                blk_tag = want;
            }
            dr = dr && !done && !pending;
        }
        STAMP(1)

        // ---- the exact count of the slot that needs it ----
        if (pending) {
            ns_rem = (uint32_t)poisson(p_mean, p_u, rcp);
            tot_samp += ns_rem;
            s = 0;
        }
        STAMP(2)

        // ---- lanes holding samples (also the ones just drawn): the sample's pair, list entries
        // ---- requested ----
        gA = c1 * B + b1; gB = c2 * B + b2; same = (gA == gB);
        if (role_e && own && !done && ns_rem > 0) {
            if (SIMPLE && !any_wide) {
                eA = ent[(b1 - blo) * 64 + lane];
                eB = same ? eA : ent[(b2 - blo) * 64 + lane];
            } else if (LISTED && have_ent) {
                // (the slot was drawn in this pass: its lists' lengths and offsets came with its entry)
                eA = pe_A; eB = same ? eA : pe_B;
            } else {
                eA = ent_get(c1, b1);
                eB = same ? eA : ent_get(c2, b2);
            }
            have_ent = false;
            const int nA = (int)(eA & 0xFFFFu), nB = (int)(eB & 0xFFFFu);
            if (nA < 1 || nB < 1 || (same && nA < 2)) {
                ns_rem = 0; // a list ran out: the slot ends (the lane draws in the next pass)
            } else {
                uint32_t w[4];
                rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB, SITE_CAND, s, w);
                i1 = (int)index_of(w[0], (uint32_t)nA);
                if (same) {
                    i2 = (int)index_of(w[1], (uint32_t)(nA - 1));
                    if (i2 >= i1) i2++;
                } else {
                    i2 = (int)index_of(w[1], (uint32_t)nB);
                }
                u_acc = u53(w[2], w[3]);
                if (n_app == 0) { // every entry sits in perm
#if defined(PMCX_PROTO_NOPERM)
                    // TIMING BUILD, WRONG RESULTS (scripts/gpu_proto.sh): a list position taken as the
                    // record's own place -- what records laid out in bin order would give the sampler:
                    // no list entry to load, one dependent round trip a sample less
                    p1 = min((eA >> 16) + (uint32_t)i1, (uint32_t)st.cap - 1u);
                    p2 = min((eB >> 16) + (uint32_t)i2, (uint32_t)st.cap - 1u);
                    last1 = min((eA >> 16) + (uint32_t)nA - 1u, (uint32_t)st.cap - 1u);
                    last2 = min((eB >> 16) + (uint32_t)nB - 1u, (uint32_t)st.cap - 1u);
                    last1b = last1;
#elif defined(PMCX_PROTO_NOLAST)
                    // TIMING BUILD, WRONG RESULTS: the lists' ends not requested with the pair (what an
                    // event that fills its hole a pass later would save)
                    p1 = perm[(eA >> 16) + i1];
                    p2 = perm[(eB >> 16) + i2];
                    last1 = p1; last2 = p2; last1b = p1;
#else
                    p1 = perm[(eA >> 16) + i1];
                    p2 = perm[(eB >> 16) + i2];
                    last1 = perm[(eA >> 16) + nA - 1];
                    last2 = perm[(eB >> 16) + nB - 1];
                    // (what the list's end will hold once one of the pair is gone: asked for
                    // here, with the others, an event does not wait for memory half-way)
                    if (same) last1b = perm[(eA >> 16) + nA - 2];
#endif
                } else {
                    p1 = list_get(gA, eA, i1, n_app);
                    p2 = list_get(gB, eB, i2, n_app);
                    last1 = list_get(gA, eA, nA - 1, n_app);
                    last2 = list_get(gB, eB, nB - 1, n_app);
                }
                evalid = true;
            }
        }
        STAMP(3)

        // ---- volumes, masses and what a coagulation will need ----
        if (evalid) {
            // the record's two 16-byte halves: everything a pair evaluation and a coagulation
            // need of a particle sits in one 32-byte record, one line
            const double2 a1 = *(const double2 *)&hot[p1].vtot, a2 = *(const double2 *)&hot[p2].vtot;
            const double2 t1 = *(const double2 *)&hot[p1].lct, t2 = *(const double2 *)&hot[p2].lct;
            v1 = a1.x; m1 = a1.y; v2 = a2.x; m2 = a2.y;
            lct1 = t1.x; gct1 = t1.y; lct2 = t2.x; gct2 = t2.y;
            // (the species volumes are summed behind the sampler, from the log: k_vol_merge)
            if (NS > 0) { no1 = cold[p1].n_orig0; no2 = cold[p2].n_orig0; }
        }
        STAMP(4)

        // ---- (5) evaluating lanes: kernel, accept ----
        if (evalid) {
            const double t_u = u_acc * k_max;
            bool need64 = true;
            if (KT == KERNEL_BROWN) {
                // single-precision screen (pmcx_device_math.h): decides the test when the
                // fp64 kernel cannot decide it differently
                EnvF32 ef;
                ef.gasfreepath = (float)envs[0 * 64 + lane]; ef.inv_gfp = (float)envs[1 * 64 + lane];
                ef.c_diff = (float)envs[2 * 64 + lane]; ef.c_speed = (float)envs[3 * 64 + lane];
                ef.c_inv_speed = (float)envs[4 * 64 + lane];
                bool ok;
                const float k32 = brown_screen_f32(v1, m1, v2, m2, ef, ok);
                const double kf32 = (double)k32 * factor;
                const double lo = kf32 * (1.0 - (double)BROWN_SCREEN_MARGIN);
                const double hi = kf32 * (1.0 + (double)BROWN_SCREEN_MARGIN);
                if (ok && k32 > 1.0e-30f && hi <= k_max && (t_u < lo || !(t_u < hi))) {
                    need64 = false;
                    acc = t_u < lo;
                }
            }
#if defined(PMCX_PROTO_ACCEPT)
            need64 = false;   // (TIMING BUILD: wrong picks give volumes the fp64 kernel was never meant for)
#endif
            if (need64) {
                EnvDerived env;
                env.gasfreepath = envs[0 * 64 + lane]; env.inv_gfp = envs[1 * 64 + lane];
                env.c_diff = envs[2 * 64 + lane]; env.c_speed = envs[3 * 64 + lane];
                env.c_inv_speed = envs[4 * 64 + lane]; env.c_fall = envs[5 * 64 + lane];
                env.rhoair = envs[6 * 64 + lane];
                const PartProps qa = props_for(ktype, v1, m1, env);
                const PartProps qb = props_for(ktype, v2, m2, env);
                const double k = kernel_pair(ktype, v1, v2, qa, qb, kp);
                const double kf = k * factor;
                if (kf > k_max) status |= ST_KMAX_EXCEEDED;
                acc = t_u < kf;
            }
#if defined(PMCX_PROTO_ACCEPT)
            // TIMING BUILD, WRONG RESULTS: the accept decision from the uniform alone at the benchmark's
            // rate (0.785 of the candidates), so that builds that pick other records run the same
            // number of events
            acc = u_acc < 0.785;
#endif
            s_ev = s;
            s++;
            ns_rem--;
        }
        STAMP(5)
        STAMP(6)

        // ---- (7) coagulate ----
        // One class, every list inside the window and in perm, room behind the new bin's list:
        // the event in straight-line code (no weighted removal, no table of entries past a
        // group's room, no look outside the window).  A lane that misses a condition takes the
        // general form below.
        bool gen = acc;
        // Nothing is in flight here on the usual path (the records have just arrived).  Saying so
        // (vmcnt(0), free there) tells the compiler that the block's bounds -- requested during
        // the draw -- are in their registers on EVERY path: otherwise the next pass's first
        // settle waits for them with the only means it has, vmcnt(0), i.e. for the event's stores
        // below, a round trip per pass in front of arithmetic that does not need it.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        lane_c = lane; c_c = c;
        asm volatile("" : "+v"(lane_c), "+v"(c_c));
        if (SIMPLE && !any_wide && __ballot(n_app != 0) == 0) {
            if (acc) {
                const double vn = v1 + v2, mn = m1 + m2;
                const bool big1 = (v1 >= v2);
                const uint32_t pn = big1 ? p1 : p2, po = big1 ? p2 : p1;
                int in = big1 ? i1 : i2;
                const int io = big1 ? i2 : i1;
                const int b_old = big1 ? b1 : b2, b_o = big1 ? b2 : b1;
                const uint32_t e_o = big1 ? eB : eA;
#if defined(PMCX_PROTO_ACCEPT)
                // (the merged particle moves up a bin in one event of seven, whatever was picked)
                const int b_new = min(b_old + ((__double_as_longlong(u_acc) & 1023) < 146 ? 1 : 0), B - 2);
                const bool move = b_new != b_old;
                const bool in_grid = true;
#else
                const int b_new = bin_index(edge, B, vn, st);
                const bool move = b_new != b_old;
                const bool in_grid = !(vn < edge[0] || vn >= edge[B]);
#endif
                // (the new bin and the one behind it inside the window: their entries are in LDS)
                bool ok = n_ev < st.max_events && in_grid &&
                          (!move || ((unsigned)(b_new - blo) < (unsigned)(SW - 1) && b_new + 1 < B));
                uint32_t en = 0;
                if (ok && move) {
                    en = ent[(b_new - blo) * 64 + lane];
                    const uint32_t nxt = ent[(b_new + 1 - blo) * 64 + lane];
                    ok = (int)(en >> 16) + (int)(en & 0xFFFFu) < (int)(nxt >> 16);
                }
                if (ok) {
                    gen = false;
                    if (b_new > (KT == KERNEL_BROWN ? (int)envs[6 * 64 + lane] : st.kmax_rng[4 * (int64_t)c + 1]))
                        status |= ST_KMAX_RANGE; // no bound was computed there
                    n_ev++;
                    if (st.mlog)
                        st.mlog[(int64_t)c * st.max_events + n_ev - 1] = (uint64_t)pn | ((uint64_t)po << 16) | ((uint64_t)big1 << 32);
                    // swap-with-last removal of the other particle
                    const int last_o = (int)(e_o & 0xFFFFu) - 1;
                    if (io != last_o) perm[(e_o >> 16) + io] = big1 ? last2 : last1;
                    cnt_set(0, b_o, e_o, last_o);
                    if (same && in == last_o) in = io;
                    if (move) {
                        // the merged particle leaves its list (look again when that list just changed)
                        uint32_t e = big1 ? eA : eB;
                        uint32_t lastv = big1 ? last1 : last2;
                        if (same) { // (the list as the removal above left it)
                            e = (e_o & 0xFFFF0000u) | (uint32_t)last_o;
                            lastv = (io == last_o - 1) ? last1 : last1b;
                        }
                        const int last = (int)(e & 0xFFFFu) - 1;
                        if (in != last) perm[(e >> 16) + in] = lastv;
                        cnt_set(0, b_old, e, last);
                        // ... and joins the list of its new bin, behind it in perm
                        const int n = (int)(en & 0xFFFFu);
                        perm[(en >> 16) + n] = pn;
                        cnt_set(0, b_new, en, n + 1);
                        PMCX_LOG(pn, b_new)
                    }
                    PMCX_LOG(po, 0xFFFFu)
                    *(double2 *)&hot[pn].vtot = make_double2(vn, mn);
                    {   // (the record's other half in one store, when either time changes)
                        const double lo_n = big1 ? lct1 : lct2, lo_o = big1 ? lct2 : lct1;
                        const double go_n = big1 ? gct1 : gct2, go_o = big1 ? gct2 : gct1;
                        if (lo_o < lo_n || go_o > go_n)
                            *(double2 *)&hot[pn].lct = make_double2(lo_o < lo_n ? lo_o : lo_n, go_o > go_n ? go_o : go_n);
                        if (NS > 0) cold[pn].n_orig0 = (int32_t)((uint32_t)no1 + (uint32_t)no2);
                    }
                    for (int sp = 1; sp < NS; sp++) {
                        int32_t *a = &norig(st, cb, sp, pn);
                        *a = (int32_t)((uint32_t)*a + (uint32_t)norig(st, cb, sp, po));
                    }
                    hot[po].vtot = -1.0; // tombstone: in the record just read
                    n_dead++;
                    tot_coag++;
                    if (PRE && move) ev_bnew = b_new;
                }
            }
        }
        if (gen) {
            do {
                if (n_ev >= st.max_events) { status |= ST_EVENT_OVERFLOW; done = true; break; }
                bool rem1 = true, rem2 = true;
                if (n1 != n2) {
                    uint32_t w[4];
                    rng_block(seed, cell_id, step, ((uint32_t)gA << 16) | (uint32_t)gB, SITE_COAG,
                              s_ev, w);
                    rem1 = u53(w[0], w[1]) < nmin / n1;
                    rem2 = u53(w[2], w[3]) < nmin / n2;
                }
                const double vn = v1 + v2, mn = m1 + m2;
                const bool big1 = (v1 >= v2);
                const bool pn_is_1 = (rem1 && rem2) ? big1 : rem1;
                const uint32_t pn = pn_is_1 ? p1 : p2, po = pn_is_1 ? p2 : p1;
                const bool rem_o = pn_is_1 ? rem2 : rem1;
                int in = pn_is_1 ? i1 : i2;
                const int io = pn_is_1 ? i2 : i1;
                // the merged particle's list (old) and the other particle's list (o)
                const int cl_old = pn_is_1 ? c1 : c2, b_old = pn_is_1 ? b1 : b2;
                const int cl_o = pn_is_1 ? c2 : c1, b_o = pn_is_1 ? b2 : b1;
                const int g_old = cl_old * B + b_old, g_o = cl_o * B + b_o;
                if (vn < edge[0] || vn >= edge[B]) status |= ST_OUT_OF_GRID;
                const int b_new = bin_index(edge, B, vn, st);
                if (b_new > (KT == KERNEL_BROWN ? (int)envs[6 * 64 + lane] : st.kmax_rng[4 * (int64_t)c + 1]))
                    status |= ST_KMAX_RANGE; // no bound was computed there
                const int g_new = cc * B + b_new;
                if (g_new != g_old && n_app >= st.max_appends) {
                    status |= ST_APPEND_OVERFLOW; done = true; break;
                }
                n_ev++;
                if (st.mlog)
                    st.mlog[(int64_t)c * st.max_events + n_ev - 1] = (uint64_t)pn | ((uint64_t)po << 16) | ((uint64_t)pn_is_1 << 32);
                if (rem_o) { // swap-with-last removal of the other particle
                    const uint32_t e = pn_is_1 ? eB : eA;
                    const int last = (int)(e & 0xFFFFu) - 1;
                    if (io != last) list_set(g_o, e, io, pn_is_1 ? last2 : last1, n_app);
                    if (ovf_of(g_o, n_app) > 0) {
                        O_last_app(g_o) = O_app_prev(O_last_app(g_o));
                        O_ovf_n(g_o) = (uint16_t)(O_ovf_n(g_o) - 1);
                    }
                    cnt_set(cl_o, b_o, e, last);
                    if (same && in == last) in = io;
                }
                if (g_new != g_old) {
                    // the merged particle leaves its list ...
                    uint32_t e = pn_is_1 ? eA : eB;
                    uint32_t lastv = pn_is_1 ? last1 : last2;
                    if (same && rem_o) { // the list just changed: look again
                        e = ent_get(cl_old, b_old);
                        lastv = list_get(g_old, e, (int)(e & 0xFFFFu) - 1, n_app);
                    }
                    const int last = (int)(e & 0xFFFFu) - 1;
                    if (in != last) list_set(g_old, e, in, lastv, n_app);
                    if (ovf_of(g_old, n_app) > 0) {
                        O_last_app(g_old) = O_app_prev(O_last_app(g_old));
                        O_ovf_n(g_old) = (uint16_t)(O_ovf_n(g_old) - 1);
                    }
                    cnt_set(cl_old, b_old, e, last);
                    // ... and joins the list of its new bin: behind it in perm while there is
                    // room (and nothing of the group sits in the table already), else the table
                    const uint32_t en = ent_get(cc, b_new);
                    const int n = (int)(en & 0xFFFFu), o = (int)(en >> 16);
                    if (ovf_of(g_new, n_app) == 0 && o + n < room_end(cc, b_new)) {
                        perm[o + n] = pn;
                    } else {
                        if (n_app == 0) // first use of the table in this call
                            for (int k = 0; k < G; k++) { O_ovf_n(k) = 0; O_last_app(k) = 0xFFFF; }
                        O_app_slot(n_app) = (uint16_t)pn;
                        O_app_prev(n_app) = O_last_app(g_new);
                        O_last_app(g_new) = (uint16_t)n_app;
                        O_ovf_n(g_new) = (uint16_t)(O_ovf_n(g_new) + 1);
                        n_app++;
                    }
                    cnt_set(cc, b_new, en, n + 1);
                    PMCX_LOG(pn, g_new)
                    if (PRE) ev_bnew = b_new;
                    // (LISTED: the list holds while no group is longer than the screen allowed for)
                    if (LISTED && n + 1 > (int)st.gnb[(int64_t)c * G + g_new]) pre_stop = true;
                }
                if (rem_o) PMCX_LOG(po, 0xFFFFu)
                *(double2 *)&hot[pn].vtot = make_double2(vn, mn);
                // source histogram, create times, class [aero_particle_coagulate]
                for (int sp = 1; sp < NS; sp++) {
                    int32_t *a = &norig(st, cb, sp, pn);
                    *a = (int32_t)((uint32_t)*a + (uint32_t)norig(st, cb, sp, po));
                }
                {   // create times and first source count share the particle's metadata record
                    const double lo_n = pn_is_1 ? lct1 : lct2, lo_o = pn_is_1 ? lct2 : lct1;
                    const double go_n = pn_is_1 ? gct1 : gct2, go_o = pn_is_1 ? gct2 : gct1;
                    if (lo_o < lo_n) hot[pn].lct = lo_o;
                    if (go_o > go_n) hot[pn].gct = go_o;
                    if (NS > 0) cold[pn].n_orig0 = (int32_t)((uint32_t)no1 + (uint32_t)no2);
                }
                if (C > 1) st.wclass[cb + pn] = cc;
                if (rem_o) { hot[po].vtot = -1.0; n_dead++; } // tombstone: in the record just read
                tot_coag++;
            } while (false);
        }
        // PRE: the list holds while the lists it was made for bound the present ones -- no bin has
        // taken more than PRE_M merged particles, none was born in a bin phase 0 did not walk
        if (PRE && acc) {
            bool bad = far;
            if (ev_bnew >= 0) {
                const int r = ev_bnew - blo;
                if (!((unsigned)r < 32u && ((pmask0 >> r) & 1u))) {
                    bad = true;                       // born in a bin phase 0 did not walk
                } else {
                    // (bins 24 .. 31 of a population wider than the LDS window share a counter)
                    const int rr = r < 24 ? r : 23;
                    const int sh = (rr < 12 ? rr : rr - 12) * 5;
                    const uint64_t a = (rr < 12 ? arr0 : arr1) + (1ull << sh);
                    if (rr < 12) arr0 = a; else arr1 = a;
                    if (((a >> sh) & 31u) > (uint64_t)PMCX_PRE_M) bad = true;   // its list outgrew the bound
                }
            }
            if (bad) pre_stop = true;
        }
        if (done) ns_rem = 0;
        if (SPLIT) {
            if (own) {
                sh_pos[lane] = (uint32_t)c1 | ((uint32_t)c2 << 8) | ((uint32_t)(b1 + 1) << 16) |
                               ((uint32_t)(b2 + 1) << 24);
                sh_ns[lane] = ns_rem; sh_s[lane] = s; sh_kmax[lane] = k_max;
                sh_flag[lane] = (done ? 1u : 0u) | (far ? 2u : 0u) | (wide ? 4u : 0u);
                if (role_e) sh_wmask[lane] = wmask;
            }
            __syncthreads(); // the other wavefront's lanes and list updates, before the next pass
        }
        STAMP(7)
    }

    if (live && role_d) // (the drawer counts the samples)
        st.n_samp[c] = tot_samp > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)tot_samp;
    if (live && !role_e && status) atomicOr(&st.status[c], status);
    if (live && role_e) {
        if (!LISTED)
        for (int cl = 0; cl < C; cl++)
            for (int k = 0; k < SW && blo + k < B; k++)
                gcnt_g[cl * B + blo + k] = (int32_t)(ent[(cl * SW + k) * 64 + lane] & 0xFFFFu);
        if (LISTED && unfinished) { // where the cell stands (a slot boundary), for the next round
            st.resume[4 * (int64_t)c] = b1 < 0 ? -1 : (b1 | (b2 << 8) | (c1 << 16) | (c2 << 24));
            st.resume[4 * (int64_t)c + 1] = n_app;
            st.resume[4 * (int64_t)c + 2] = 0; st.resume[4 * (int64_t)c + 3] = 0;
            next_list[1 + atomicAdd(next_list, 1)] = c;
        }
        if (n_app && !unfinished) { // (a cell handed to another launch: judged where its sweep ends)
            bool ovf = false;
            for (int g = 0; g < G; g++) ovf |= O_ovf_n(g) != 0;
            if (ovf) st.sort_valid[c] = 0;
        }
        if (n_lg >= 0) st.n_log[c] = n_lg;
        else if (n_lg == -2) st.key_valid[c] = 0;
        if (PRE && unfinished) { // where the cell stands, for the launch behind this one
            st.resume[4 * (int64_t)c] = b1 < 0 ? -1 : (b1 | (b2 << 8));
            st.resume[4 * (int64_t)c + 1] = n_app;
            st.resume[4 * (int64_t)c + 2] = 0; st.resume[4 * (int64_t)c + 3] = 0;   // (left at a slot boundary)
            resume_count[1 + atomicAdd(resume_count, 1)] = c;
        }
        st.n_events[c] = n_ev;
        st.n_part[c] = st.n_part[c] - n_dead;
        st.n_coag[c] = tot_coag;
        if (status) atomicOr(&st.status[c], status);
    }
    STAMP(8)
    if (STAMPS && lane == 0) {
        for (int k = 0; k < 11; k++) atomicAdd(&g_seq_stamps[k], (unsigned long long)t_acc[k]);
        atomicAdd(&g_seq_stamps[11], 1ull);
    }
#undef PMCX_SETTLE
#undef PMCX_LOG
#undef STAMP
#undef MASK
#undef MASKC
#undef gcnt_g
#undef goff_g
#undef O_ovf_n
#undef O_last_app
#undef O_app_slot
#undef O_app_prev
}

// ------------------------------ coag sampler, lane per cell, the cells of a workgroup grouped by what they do next
//
// k_coag_grp (one class, Brownian: the benchmark's configuration, any species count).  In
// k_coag_lane a lane IS a cell for the whole sweep, and the 64 cells of a wavefront are in different states
// -- settling slots, counting, evaluating a sample, coagulating -- so every pass runs every stage for the
// few lanes that need it: 1 700 instructions a pass, the draw alone half of a wavefront's cycles at a sixth
// of its lanes' use.  Here a workgroup of 256 threads owns 256 cells whose whole sweep state
// lives in LDS, and every round the cells are sorted by what they need: those holding samples are
// evaluated 64 to a wavefront by code that only evaluates (and coagulates), those holding none draw 64 to
// a wavefront by code that only draws -- every lane of a chunk busy with the same stage.  Per cell the
// sequence of operations is k_coag_lane's, and the oracle's.  Only the straight path is here: a cell whose
// population reaches beyond the window, or whose next event needs the general code (no room behind the new
// bin's list, a full event log, a volume outside the grid), is handed -- where it stands, in the middle of
// a slot if need be -- to k_coag_lane, launched behind this kernel over the cells listed (st.resume).
constexpr int GRP_NC = 256;                  // cells (threads) of a workgroup
constexpr int GRP_SW = 32;                   // bins of a cell's window (its list state in LDS)
#ifndef PMCX_GRP_NTRY
#define PMCX_GRP_NTRY 8
#endif
#ifndef PMCX_GRP_ROUNDS
#define PMCX_GRP_ROUNDS 3
#endif
constexpr uint32_t GRP_DONE = 1u, GRP_HAND = 2u;

__host__ __device__ inline size_t grp_lds_bytes(int B)
{
    size_t o = ((size_t)(B + 1) * sizeof(double) + 15) & ~(size_t)15;       // sort-grid edges
    o += sizeof(double) * (POISSON_RCP_N + 2);                             // 1/n of the Poisson walk
    o += sizeof(double) * 2 * GRP_NC + sizeof(uint64_t) * 4 * GRP_NC;      // k_max, weight; the cached block and bounds
    o += sizeof(uint32_t) * (size_t)GRP_SW * GRP_NC;                       // count | offset << 16 of the window's bins
    o += sizeof(uint32_t) * 18 * GRP_NC;                                   // position, samples, ..., counters (18 words)
    o += sizeof(uint16_t) * 2 * GRP_NC + sizeof(uint32_t) * 8;             // the two queues, the wavefronts' counts
    return (o + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(GRP_NC, 2) void k_coag_grp(DeviceState st, double del_t, uint64_t seed, uint32_t step,
                                                        int32_t *resume_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NC = GRP_NC, SW = GRP_SW, NTRY = PMCX_GRP_NTRY;
    constexpr uint32_t TAG_NONE = 0xFFFFFFFFu;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int B = st.n_bin, NS = st.n_source;
    const uint64_t lt = (1ull << lane) - 1ull;
    double *edge = (double *)smem;
    double *rcp = (double *)(smem + (((size_t)(B + 1) * sizeof(double) + 15) & ~(size_t)15));
    double *s_kmax = rcp + (POISSON_RCP_N + 2), *s_mag = s_kmax + NC;
    uint64_t *s_wq01 = (uint64_t *)(s_mag + NC), *s_wq23 = s_wq01 + NC, *s_kq01 = s_wq23 + NC, *s_kq23 = s_kq01 + NC;
    uint32_t *ent = (uint32_t *)(s_kq23 + NC);                       // [SW][NC]
    uint32_t *s_pos = ent + (size_t)SW * NC;                         // (b1 + 1) | (b2 + 1) << 8 | blo << 16
    uint32_t *s_flag = s_pos + NC, *s_ns = s_flag + NC, *s_s = s_ns + NC, *s_wmask = s_s + NC, *s_tag = s_wmask + NC;
    float *s_env = (float *)(s_tag + NC);                            // [5][NC]
    int32_t *s_rng = (int32_t *)(s_env + 5 * NC), *s_nev = s_rng + NC, *s_ncoag = s_nev + NC, *s_samp = s_ncoag + NC,
            *s_status = s_samp + NC, *s_nlg = s_status + NC, *s_lo = s_nlg + NC;   // (s_lo: spare)
    uint16_t *qD = (uint16_t *)(s_lo + NC), *qE = qD + NC;
    uint32_t *s_cnt = (uint32_t *)(qE + NC);                         // [8]

    for (int j = t; j <= B; j += NC) edge[j] = st.v_edge[j];
    for (int j = t; j <= POISSON_RCP_N; j += NC) rcp[j] = g_poisson_rcp[j];
#ifdef PMCX_GRP_STAMPS
    uint64_t t_last = __builtin_amdgcn_s_memtime(), t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define GSTAMP(k) { const uint64_t t_now = __builtin_amdgcn_s_memtime(); t_acc[k] += t_now - t_last; t_last = t_now; }
#else
#define GSTAMP(k)
#endif

    // ---- set-up: thread t owns cell blockIdx.x * NC + t for this part and for the write-back
    const int cidx = blockIdx.x * NC + t;
    const bool live = cidx < st.n_cells && in_loop(st, cidx);
    const int c_own = cidx < st.n_cells ? cidx : st.n_cells - 1;
    if (cidx < st.n_cells && !live) { st.n_samp[cidx] = 0; st.n_coag[cidx] = 0; st.n_events[cidx] = 0; }
    int blo_own = 0;
    {
        const int32_t *gc = st.gcnt + (int64_t)c_own * B;
        const int32_t *go = st.goff + (int64_t)c_own * (B + 1);
        uint64_t m0 = 0, m1 = 0;
        int blo = B;
        for (int b0 = 0; b0 < B; b0 += 8) {
            int32_t n8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) n8[k] = gc[min(b0 + k, B - 1)];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int b = b0 + k;
                const bool has = b < B && n8[k] > 0;
                const uint64_t bit = has ? 1ull << (b & 63) : 0ull;
                if (b < 64) m0 |= bit; else m1 |= bit;
                blo = (has && b < blo) ? b : blo;
            }
        }
        if (blo > B - SW) blo = (B - SW > 0) ? B - SW : 0;
        uint32_t e[SW];
#pragma unroll
        for (int k = 0; k < SW; k++) {
            const int b = min(blo + k, B - 1);
            e[k] = (uint32_t)gc[b] | ((uint32_t)go[b] << 16);
        }
#pragma unroll
        for (int k = 0; k < SW; k++) ent[k * NC + t] = (blo + k < B) ? e[k] : 0u;
        uint64_t w64;
        if (blo == 0) w64 = m0;
        else if (blo < 64) w64 = (m0 >> blo) | (m1 << (64 - blo));
        else w64 = m1 >> (blo - 64);
        bool far = (w64 >> 32) != 0 || (blo < 64 && (m1 >> blo) != 0);
        if (blo > 0 && blo < 64 && (m0 & ((1ull << blo) - 1ull))) far = true;
        if (blo >= 64 && (m0 || (m1 & ((1ull << (blo - 64)) - 1ull)))) far = true;
        blo_own = blo;
        uint32_t fl = live ? 0u : GRP_DONE;
        if (live && far) fl = GRP_DONE | GRP_HAND;       // (a population beyond the window: the other kernel, from the start)
        s_flag[t] = fl;
        s_wmask[t] = (uint32_t)w64;
        s_pos[t] = ((uint32_t)blo << 16);
        s_ns[t] = 0; s_s[t] = 0; s_tag[t] = TAG_NONE; s_kmax[t] = 0.0;
        s_wq01[t] = 0; s_wq23[t] = 0; s_kq01[t] = 0; s_kq23[t] = 0;
        s_mag[t] = st.mag[c_own];
        const EnvDerived ed = env_derive(st.temp[c_own], st.pressure[c_own]);
        s_env[0 * NC + t] = (float)ed.gasfreepath; s_env[1 * NC + t] = (float)ed.inv_gfp;
        s_env[2 * NC + t] = (float)ed.c_diff; s_env[3 * NC + t] = (float)ed.c_speed;
        s_env[4 * NC + t] = (float)ed.c_inv_speed;
        s_rng[t] = st.kmax_rng[4 * (int64_t)c_own + 1];
        s_nev[t] = 0; s_ncoag[t] = 0; s_samp[t] = 0; s_status[t] = 0;
        s_nlg[t] = st.key_valid[c_own] != 0 ? st.n_log[c_own] : -1;
    }
    __syncthreads();
    GSTAMP(0)

    const KernelParams kp{st.const_k, st.additive_beta};

    // ---- the draw: the cell settles slots until one has samples (or the sweep is over), a few rounds of
    // ---- NTRY tries with the fetch of the next block in between
    auto draw = [&](int k, bool valid) {
        const int c = min(blockIdx.x * NC + k, st.n_cells - 1);
        const uint32_t cell_id = cell_gid(st, c);
        const uint32_t pos = s_pos[k];
        int b1 = (int)(pos & 255u) - 1, b2 = (int)((pos >> 8) & 255u) - 1;
        const int blo = (int)(pos >> 16);
        const uint32_t wmask = s_wmask[k];
        uint32_t blk_tag = s_tag[k];
        uint64_t wq01 = s_wq01[k], wq23 = s_wq23[k], kq01 = s_kq01[k], kq23 = s_kq23[k];
        const double fmax = s_mag[k];
        const float cf_up = (float)(fmax * del_t) * 1.000001f;
        bool dr = valid, done = false, pending = false;
        double p_mean = 0, p_u = 0, k_max = 0;
        for (int r = 0; r < PMCX_GRP_ROUNDS; r++) {
            if (__ballot(dr) == 0) break;
            uint32_t want = TAG_NONE;
            bool active = dr;
#pragma unroll
            for (int q = 0; q < NTRY; q++) {
                if (active) {
                    const int p1_ = b1 - blo, p2_ = b2 - blo;
                    const bool rowok = b1 >= 0 && ((wmask >> p1_) & 1u);
                    const uint32_t m = (rowok && p2_ < 31) ? (wmask & (~0u << (p2_ + 1))) : 0u;
                    const uint32_t mr = b1 < 0 ? wmask : (p1_ < 31 ? (wmask & (~0u << (p1_ + 1))) : 0u);
                    const int nb1 = m ? b1 : blo + __builtin_ctz(mr | 0x80000000u);
                    const int nb2 = m ? blo + __builtin_ctz(m) : nb1;
                    if (!m && !mr) { done = true; active = false; }
                    else {
                        const uint32_t tag = ((uint32_t)nb1 << 16) | ((uint32_t)nb2 & ~3u);
                        if (tag != blk_tag) { want = tag; active = false; }
                        else {
                            b1 = nb1; b2 = nb2;
                            const uint32_t dA = ent[(nb1 - blo) * NC + k], dB = ent[(nb2 - blo) * NC + k];
                            const uint32_t hB_ = (uint32_t)nb2, sh_ = (hB_ & 1u) * 32u;
                            const uint32_t wu_ = (uint32_t)(((hB_ & 2u) ? wq23 : wq01) >> sh_);
                            const float kf_ = __uint_as_float((uint32_t)(((hB_ & 2u) ? kq23 : kq01) >> sh_));
                            const uint32_t nAu = dA & 0xFFFFu, nBu = dB & 0xFFFFu;
                            const uint32_t np_ = nb1 == nb2 ? (nAu * (nAu - 1u)) >> 1 : nAu * nBu;
                            const float mean_up = (kf_ * cf_up) * ((float)np_ * 1.00001f);
                            const float u_up = (float)((wu_ >> 8) + 1u) * 0x1.0p-24f;
                            bool zero = u_up < (1.0f - mean_up) - 1e-6f;
                            if (!zero) {
                                const double nA_ = (double)nAu, nB_ = (double)nBu;
                                k_max = (double)kf_ * fmax;
                                const double n_possible_ = (nb1 == nb2) ? nA_ * (nA_ - 1.0) / 2.0 : nA_ * nB_;
                                const double mean_ = k_max * del_t * n_possible_;
                                const double u_ = (double)wu_ * 0x1.0p-32;
                                zero = u_ < (1.0 - mean_) - 4e-15;
                                if (!zero) { pending = true; p_mean = mean_; p_u = u_; active = false; }
                            }
                        }
                    }
                }
            }
            if (want != TAG_NONE) { // the block the sweep moves into: one Philox block, four bounds
                uint32_t w[4];
                rng_block(seed, cell_id, step, want, SITE_POISSON, 0, w);
                wq01 = (uint64_t)w[0] | ((uint64_t)w[1] << 32); wq23 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
                const int hA = (int)(want >> 16), hBq = (int)(want & 0xFFFFu);
                const float *row = st.kmax + (int64_t)c * st.kmax_stride + hA * B;
                const float kq0 = row[min(hBq, B - 1)], kq1 = row[min(hBq + 1, B - 1)];
                const float kq2 = row[min(hBq + 2, B - 1)], kq3 = row[min(hBq + 3, B - 1)];
                kq01 = (uint64_t)__float_as_uint(kq0) | ((uint64_t)__float_as_uint(kq1) << 32);
                kq23 = (uint64_t)__float_as_uint(kq2) | ((uint64_t)__float_as_uint(kq3) << 32);
                blk_tag = want;
            }
            dr = dr && !done && !pending;
        }
        if (valid) {
            if (pending) {
                const uint32_t ns = (uint32_t)poisson(p_mean, p_u, rcp);
                s_ns[k] = ns; s_s[k] = 0; s_kmax[k] = k_max;
                s_samp[k] = (int32_t)((uint32_t)s_samp[k] + ns);
            }
            if (done) s_flag[k] = s_flag[k] | GRP_DONE;
            s_pos[k] = (uint32_t)(b1 + 1) | ((uint32_t)(b2 + 1) << 8) | ((uint32_t)blo << 16);
            s_tag[k] = blk_tag; s_wq01[k] = wq01; s_wq23[k] = wq23; s_kq01[k] = kq01; s_kq23[k] = kq23;
        }
    };

    // ---- one sample of the cell's slot: the pair, its records, kernel and accept, the coagulation
    auto eval = [&](int k, bool valid) {
        const int c = min(blockIdx.x * NC + k, st.n_cells - 1);
        const int64_t cb = (int64_t)c * st.cap;
        const uint32_t cell_id = cell_gid(st, c);
        uint16_t *perm = st.perm + (int64_t)c * st.pcap;
        Hot *hot = st.hot + cb;
        Cold *cold = st.cold + cb;
        const uint32_t pos = s_pos[k];
        const int b1 = (int)(pos & 255u) - 1, b2 = (int)((pos >> 8) & 255u) - 1, blo = (int)(pos >> 16);
        uint32_t ns_rem = valid ? s_ns[k] : 0u, s = s_s[k];
        const double k_max = s_kmax[k], factor = s_mag[k];
        uint32_t wmask = s_wmask[k];
        int n_ev = s_nev[k], n_lg = s_nlg[k], status = 0, tot_coag = s_ncoag[k];
        const bool same = b1 == b2;
        bool evalid = false, acc = false, hand = false;
        int i1 = 0, i2 = 0;
        uint32_t p1 = 0, p2 = 0, last1 = 0, last2 = 0, last1b = 0, eA = 0, eB = 0;
        double u_acc = 0;
        if (ns_rem > 0) {
            eA = ent[(b1 - blo) * NC + k];
            eB = same ? eA : ent[(b2 - blo) * NC + k];
            const int nA = (int)(eA & 0xFFFFu), nB = (int)(eB & 0xFFFFu);
            if (nA < 1 || nB < 1 || (same && nA < 2)) {
                ns_rem = 0; // a list ran out: the slot ends
            } else {
                uint32_t w[4];
                rng_block(seed, cell_id, step, ((uint32_t)b1 << 16) | (uint32_t)b2, SITE_CAND, s, w);
                i1 = (int)index_of(w[0], (uint32_t)nA);
                if (same) {
                    i2 = (int)index_of(w[1], (uint32_t)(nA - 1));
                    if (i2 >= i1) i2++;
                } else {
                    i2 = (int)index_of(w[1], (uint32_t)nB);
                }
                u_acc = u53(w[2], w[3]);
                p1 = perm[(eA >> 16) + i1];
                p2 = perm[(eB >> 16) + i2];
                last1 = perm[(eA >> 16) + nA - 1];
                last2 = perm[(eB >> 16) + nB - 1];
                if (same) last1b = perm[(eA >> 16) + nA - 2];
                evalid = true;
            }
        }
        double v1 = 0, v2 = 0, m1 = 0, m2 = 0, lct1 = 0, lct2 = 0, gct1 = 0, gct2 = 0;
        int32_t no1 = 0, no2 = 0;
        if (evalid) {
            const double2 a1 = *(const double2 *)&hot[p1].vtot, a2 = *(const double2 *)&hot[p2].vtot;
            const double2 t1 = *(const double2 *)&hot[p1].lct, t2 = *(const double2 *)&hot[p2].lct;
            v1 = a1.x; m1 = a1.y; v2 = a2.x; m2 = a2.y;
            lct1 = t1.x; gct1 = t1.y; lct2 = t2.x; gct2 = t2.y;
            if (NS > 0) { no1 = cold[p1].n_orig0; no2 = cold[p2].n_orig0; }
        }
        if (evalid) {
            const double t_u = u_acc * k_max;
            bool need64 = true;
            {
                EnvF32 ef;
                ef.gasfreepath = s_env[0 * NC + k]; ef.inv_gfp = s_env[1 * NC + k];
                ef.c_diff = s_env[2 * NC + k]; ef.c_speed = s_env[3 * NC + k]; ef.c_inv_speed = s_env[4 * NC + k];
                bool ok;
                const float k32 = brown_screen_f32(v1, m1, v2, m2, ef, ok);
                const double kf32 = (double)k32 * factor;
                const double lo = kf32 * (1.0 - (double)BROWN_SCREEN_MARGIN);
                const double hi = kf32 * (1.0 + (double)BROWN_SCREEN_MARGIN);
                if (ok && k32 > 1.0e-30f && hi <= k_max && (t_u < lo || !(t_u < hi))) {
                    need64 = false;
                    acc = t_u < lo;
                }
            }
            if (need64) { // (some lane within a thousandth of the accept line: a few chunks in a hundred)
                const EnvDerived env = env_derive(st.temp[c], st.pressure[c]);
                const PartProps qa = props_for(KERNEL_BROWN, v1, m1, env);
                const PartProps qb = props_for(KERNEL_BROWN, v2, m2, env);
                const double kk = kernel_pair(KERNEL_BROWN, v1, v2, qa, qb, kp);
                const double kf = kk * factor;
                if (kf > k_max) status |= ST_KMAX_EXCEEDED;
                acc = t_u < kf;
            }
        }
        if (evalid && acc) {
            const double vn = v1 + v2, mn = m1 + m2;
            const bool big1 = (v1 >= v2);
            const uint32_t pn = big1 ? p1 : p2, po = big1 ? p2 : p1;
            int in = big1 ? i1 : i2;
            const int io = big1 ? i2 : i1;
            const int b_old = big1 ? b1 : b2, b_o = big1 ? b2 : b1;
            const uint32_t e_o = big1 ? eB : eA;
            const int b_new = bin_index(edge, B, vn, st);
            const bool move = b_new != b_old;
            const bool in_grid = !(vn < edge[0] || vn >= edge[B]);
            bool ok = n_ev < st.max_events && in_grid &&
                      (!move || ((unsigned)(b_new - blo) < (unsigned)(SW - 1) && b_new + 1 < B));
            uint32_t en = 0;
            if (ok && move) {
                en = ent[(b_new - blo) * NC + k];
                const uint32_t nxt = ent[(b_new + 1 - blo) * NC + k];
                ok = (int)(en >> 16) + (int)(en & 0xFFFFu) < (int)(nxt >> 16);
            }
            if (!ok) {
                hand = true;          // the general event: the other kernel, from this very sample
            } else {
                auto cnt_set = [&](int b, uint32_t e, int n) {
                    ent[(b - blo) * NC + k] = (e & 0xFFFF0000u) | (uint32_t)n;
                    if (n == 0) wmask &= ~(1u << (b - blo));
                    else wmask |= 1u << (b - blo);
                };
                auto log_key = [&](uint32_t slot, uint32_t key) {
                    if (n_lg >= 0) {
                        if (n_lg < EVLOG_CAP) st.evlog[(int64_t)c * EVLOG_CAP + n_lg++] = slot | (key << 16);
                        else n_lg = -2;
                    }
                };
                if (b_new > s_rng[k]) status |= ST_KMAX_RANGE; // no bound was computed there
                n_ev++;
                if (st.mlog)
                    st.mlog[(int64_t)c * st.max_events + n_ev - 1] = (uint64_t)pn | ((uint64_t)po << 16) | ((uint64_t)big1 << 32);
                // swap-with-last removal of the other particle
                const int last_o = (int)(e_o & 0xFFFFu) - 1;
                if (io != last_o) perm[(e_o >> 16) + io] = big1 ? last2 : last1;
                cnt_set(b_o, e_o, last_o);
                if (same && in == last_o) in = io;
                if (move) {
                    uint32_t e = big1 ? eA : eB;
                    uint32_t lastv = big1 ? last1 : last2;
                    if (same) { // (the list as the removal above left it)
                        e = (e_o & 0xFFFF0000u) | (uint32_t)last_o;
                        lastv = (io == last_o - 1) ? last1 : last1b;
                    }
                    const int last = (int)(e & 0xFFFFu) - 1;
                    if (in != last) perm[(e >> 16) + in] = lastv;
                    cnt_set(b_old, e, last);
                    const int n = (int)(en & 0xFFFFu);
                    perm[(en >> 16) + n] = pn;
                    cnt_set(b_new, en, n + 1);
                    log_key(pn, (uint32_t)b_new);
                }
                log_key(po, 0xFFFFu);
                *(double2 *)&hot[pn].vtot = make_double2(vn, mn);
                {
                    const double lo_n = big1 ? lct1 : lct2, lo_o = big1 ? lct2 : lct1;
                    const double go_n = big1 ? gct1 : gct2, go_o = big1 ? gct2 : gct1;
                    if (lo_o < lo_n || go_o > go_n)
                        *(double2 *)&hot[pn].lct = make_double2(lo_o < lo_n ? lo_o : lo_n, go_o > go_n ? go_o : go_n);
                    if (NS > 0) cold[pn].n_orig0 = (int32_t)((uint32_t)no1 + (uint32_t)no2);
                }
                for (int sp = 1; sp < NS; sp++) {
                    int32_t *a = &norig(st, cb, sp, pn);
                    *a = (int32_t)((uint32_t)*a + (uint32_t)norig(st, cb, sp, po));
                }
                hot[po].vtot = -1.0; // tombstone: in the record just read
                tot_coag++;
            }
        }
        if (valid) {
            if (hand) {
                s_flag[k] = s_flag[k] | GRP_DONE | GRP_HAND;     // (s, ns_rem as they stand: this sample is the other kernel's)
            } else if (evalid) {
                s++; ns_rem--;
            }
            s_ns[k] = hand ? ns_rem : ns_rem; s_s[k] = s;
            s_wmask[k] = wmask; s_nev[k] = n_ev; s_nlg[k] = n_lg; s_ncoag[k] = tot_coag;
            if (status) s_status[k] = s_status[k] | status;
        }
    };

    // ---- rounds: the workgroup's cells sorted by what they do next, 64 of a kind to a wavefront
    while (true) {
        const uint32_t fl = s_flag[t], nsv = s_ns[t];
        const bool wE = !(fl & GRP_DONE) && nsv > 0, wD = !(fl & GRP_DONE) && nsv == 0;
        const uint64_t mE = __ballot(wE), mD = __ballot(wD);
        if (lane == 0) { s_cnt[wave] = (uint32_t)__popcll(mD); s_cnt[4 + wave] = (uint32_t)__popcll(mE); }
        __syncthreads();
        int baseD = 0, baseE = 0, nD = 0, nE = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) { baseD += (int)s_cnt[w]; baseE += (int)s_cnt[4 + w]; }
            nD += (int)s_cnt[w]; nE += (int)s_cnt[4 + w];
        }
        if (wD) qD[baseD + __popcll(mD & lt)] = (uint16_t)t;
        if (wE) qE[baseE + __popcll(mE & lt)] = (uint16_t)t;
        __syncthreads();
        GSTAMP(1)
        if (nD + nE == 0) break;
        const int chE = (nE + 63) >> 6, chD = (nD + 63) >> 6;
        for (int ch = wave; ch < chE + chD; ch += 4) {
            if (ch < chE) {
                const int idx = ch * 64 + lane;
                eval(idx < nE ? (int)qE[idx] : 0, idx < nE);
                GSTAMP(2)
#ifdef PMCX_GRP_STAMPS
                t_acc[6]++;
#endif
            } else {
                const int idx = (ch - chE) * 64 + lane;
                draw(idx < nD ? (int)qD[idx] : 0, idx < nD);
                GSTAMP(3)
#ifdef PMCX_GRP_STAMPS
                t_acc[7]++;
#endif
            }
        }
        __syncthreads();
        GSTAMP(4)
#ifdef PMCX_GRP_STAMPS
        t_acc[10]++;
#endif
    }

    // ---- write-back (thread t: its own cell)
    if (live) {
        const int c = cidx;
        const uint32_t fl = s_flag[t];
        int32_t *gc = st.gcnt + (int64_t)c * B;
        for (int k = 0; k < SW && blo_own + k < B; k++) gc[blo_own + k] = (int32_t)(ent[k * NC + t] & 0xFFFFu);
        const int n_lg = s_nlg[t];
        if (n_lg >= 0) st.n_log[c] = n_lg;
        else if (n_lg == -2) st.key_valid[c] = 0;
        st.n_samp[c] = s_samp[t];
        st.n_events[c] = s_nev[t];
        st.n_part[c] = st.n_part[c] - s_ncoag[t];
        st.n_coag[c] = s_ncoag[t];
        if (s_status[t]) atomicOr(&st.status[c], s_status[t]);
        if (fl & GRP_HAND) { // where the cell stands, for the kernel launched behind this one
            const uint32_t pos = s_pos[t];
            const int b1 = (int)(pos & 255u) - 1, b2 = (int)((pos >> 8) & 255u) - 1;
            st.resume[4 * (int64_t)c] = b1 < 0 ? -1 : (b1 | (b2 << 8));
            st.resume[4 * (int64_t)c + 1] = 0;
            st.resume[4 * (int64_t)c + 2] = (int32_t)s_ns[t];
            st.resume[4 * (int64_t)c + 3] = (int32_t)s_s[t];
            resume_count[1 + atomicAdd(resume_count, 1)] = c;
        }
    }
#ifdef PMCX_GRP_STAMPS
    GSTAMP(5)
    if (lane == 0) {
        for (int k = 0; k < 11; k++) atomicAdd(&g_seq_stamps[k], (unsigned long long)t_acc[k]);
        atomicAdd(&g_seq_stamps[11], 1ull);
    }
#endif
#undef GSTAMP
}

hipError_t seq_stamps_read(uint64_t out[12], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seq_stamps), 12 * sizeof(uint64_t));
    if (e == hipSuccess && reset) {
        const uint64_t z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_seq_stamps), z, sizeof(z));
    }
    return e;
}

// ------------------------------------- coag sampler, several weight classes: the screen
//
// With C weight classes the sweep visits C (C + 1) / 2 class pairs times the bin pairs of each
// (WRFV3/Registry/registry.partmc_trans:2-41 carries 40 classes: 5 * 10^6 slots a cell), the events stay
// what they were, and nearly every slot draws a count of zero -- but each needs its own uniform to
// say so.  A lane walking its cell's slots one Philox block after the other (the forms above) takes
// 67 ms at 10 classes and 1.8 s at 40 for the 75 000 cells it does in 1.2 ms with one class.  Hence, as in
// the one-class form with the draw split off (PRE), but with the whole wavefront on one cell: the
// slots of a class pair are laid side by side over the lanes, four partner bins (one Philox block) to
// a lane, tested against 1 - mean with every list a few entries longer than it is now, and the few
// that fail are listed in sweep order {slot, uniform}.  k_coag_lane<.., LISTED> then sweeps that
// list, a lane per cell, taking each entry's exact count from the lists as they are when it reaches
// it: every count is the oracle's.  Walked: per class the bins from its first populated one up to two
// above the cell's last (a merged particle joins the class of one of its parents, at or above that
// parent's bin); gnb holds the length each group was tested with, and the sweep stops at the first
// arrival that exceeds one (or when the list ran full): the next round screens what lies behind the
// slot it settled last, with the lists as they are then (launch_coag_listed).
// the length a group is tested with: n + max(SCR_M, n >> SCR_SHIFT) for a populated one, SCR_E for an
// empty one that could take a merged particle.  Measured at 75 000 cells (ms per mc_coag, 2 / 10 / 40
// classes; profiles/r4_classes.txt): 12, -, 12 with 512 entries a list: 3.2 / 10.2 / 90.9 (14 rounds at 40
// classes, every list full); with 2048 entries: 3.2 / 9.1 / 76.9; 4, -, 2: 4.3 / 7.6 / 54.8; 2, 2, 1: 3.0 /
// 7.9 / 55.6; 3, 3, 3: 2.9 / 7.7 / 56.5 (2.5 rounds at 40 classes)
#ifndef PMCX_SCR_M
#define PMCX_SCR_M 3
#endif
#ifndef PMCX_SCR_SHIFT
#define PMCX_SCR_SHIFT 3
#endif
#ifndef PMCX_SCR_E
#define PMCX_SCR_E 3
#endif
constexpr int SCR_M = PMCX_SCR_M, SCR_SHIFT = PMCX_SCR_SHIFT, SCR_E = PMCX_SCR_E;

__host__ __device__ inline size_t screen_lds_per_wave(int G, int C)
{
    return (((size_t)2 * G + 15) & ~(size_t)15) + (((size_t)10 * C + 15) & ~(size_t)15);
}

__global__ __launch_bounds__(256) void k_coag_screen(DeviceState st, double del_t, uint64_t seed, uint32_t step,
                                                     const int32_t *list, const int32_t *count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = st.n_bin, C = st.n_class, G = B * C;
    unsigned char *mine = smem + (size_t)wave * screen_lds_per_wave(G, C);
    uint16_t *nup = (uint16_t *)mine;                                        // [G] the length a group is tested with
    double *mg = (double *)(mine + (((size_t)2 * G + 15) & ~(size_t)15));    // [C] weights
    int16_t *lo = (int16_t *)(mg + C);                                       // [C] first populated bin (B: none)
    const int n_work = list ? *count : st.n_cells;
    for (int wk = blockIdx.x * 4 + wave; wk < n_work; wk += gridDim.x * 4) {
        const int c = list ? list[wk] : wk;
        if (!in_loop(st, c)) continue;
        // settled so far: everything up to and including the slot (sc1, sc2, sb1, sb2); sc1 < 0: nothing
        int sc1 = -1, sc2 = 0, sb1 = 0, sb2 = 0;
        if (list) {
            const int32_t pos = st.resume[4 * (int64_t)c];
            if (pos != -1) { sb1 = pos & 0xFF; sb2 = (pos >> 8) & 0xFF; sc1 = (pos >> 16) & 0xFF; sc2 = (pos >> 24) & 0xFF; }
        }
        const int32_t *gc = st.gcnt + (int64_t)c * G;
        uint16_t *nb_g = st.gnb + (int64_t)c * G;
        uint2 *sl = st.slist + (int64_t)c * SLIST_CAP;
        const float *kmx = st.kmax + (int64_t)c * st.kmax_stride;
        const uint32_t cell_id = cell_gid(st, c);
        int hi = -1;
        for (int g = lane; g < G; g += 64) {
            const int n = gc[g];
            nup[g] = (uint16_t)n;
            if (n > 0) hi = max(hi, g % B);
        }
        for (int d = 32; d >= 1; d >>= 1) hi = max(hi, __shfl_xor(hi, d, 64));
        __threadfence_block();
        for (int cl = lane; cl < C; cl += 64) {
            int l = B;
            for (int b = B - 1; b >= 0; b--) if (nup[cl * B + b]) l = b;
            lo[cl] = (int16_t)l;
            mg[cl] = st.mag[(int64_t)c * C + cl];
        }
        __threadfence_block();
        // (the cell's bounds cover the populated range of the call's start and bounds_margin >= 2 bins above it)
        const int hi2 = min(min(B - 1, hi + 2), st.kmax_rng[4 * (int64_t)c + 1]);
        for (int g = lane; g < G; g += 64) {
            const int cl = g / B, b = g - cl * B, l = lo[cl];
            const bool walk = l < B && b >= l && b <= hi2;
            const uint32_t n = nup[g];
            const uint32_t v = !walk ? 0u : (n == 0u ? (uint32_t)SCR_E : min(65535u, n + max((uint32_t)SCR_M, n >> SCR_SHIFT)));
            nup[g] = (uint16_t)v;
            nb_g[g] = (uint16_t)v;
        }
        __threadfence_block();

        int npre = 0;
        bool full = false;
        // (diagnostic, pmcx_set_sampler_variant(5, 1 | 2, 0): lists of 64 / 256 entries, so that tests run many rounds)
        const int cap_l = st.force_lanes == 5 && st.force_wpb == 1 ? 64 : (st.force_lanes == 5 && st.force_wpb == 2 ? 256 : SLIST_CAP);
        for (int c1 = max(sc1, 0); c1 < C && !full && hi >= 0; c1++) {
            const int l1 = lo[c1];
            if (l1 >= B) continue;
            const double n1 = mg[c1];
            for (int c2 = (c1 == sc1 ? sc2 : 0); c2 <= c1 && !full; c2++) {
                const int l2 = lo[c2];
                if (l2 >= B) continue;
                const double n2 = mg[c2];
                const double fmax = (n1 < n2) ? n2 : n1;
                const bool at_start = (c1 == sc1 && c2 == sc2);
                // rows l1 .. hi2 of class c1, the blocks of four partner bins from l2's up to hi2's
                const int q0 = l2 >> 2, nblk = (hi2 >> 2) - q0 + 1, items = (hi2 - l1 + 1) * nblk;
                for (int t0 = 0; t0 < items && !full; t0 += 64) {
                    const int t = t0 + lane;
                    const int r = t / nblk, q = t - r * nblk;
                    const int b1 = l1 + r, q4 = (q0 + q) << 2;
                    bool act = t < items;
                    if (c1 == c2 && q4 + 3 < b1) act = false;                                  // (partners below the row's bin)
                    if (at_start && (b1 < sb1 || (b1 == sb1 && q4 + 3 <= sb2))) act = false;  // (settled already)
                    uint32_t pm = 0;
                    uint32_t w[4] = {0, 0, 0, 0};
                    const uint32_t gA = (uint32_t)(c1 * B + b1), gB0 = (uint32_t)(c2 * B + q4);
                    if (act) {
                        rng_block(seed, cell_id, step, (gA << 16) | gB0, SITE_POISSON, 0, w);
                        const float4 kf4 = *(const float4 *)(kmx + b1 * B + q4);
                        const float kfa[4] = {kf4.x, kf4.y, kf4.z, kf4.w};
                        const double nA_up = (double)nup[gA];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int b2 = q4 + k;
                            const bool ok = b2 >= l2 && b2 <= hi2 && (c1 != c2 || b2 >= b1) &&
                                            !(at_start && b1 == sb1 && b2 <= sb2);
                            const double nB_up = (double)nup[gB0 + (ok ? k : 0)];
                            const double np_up = (c1 == c2 && b2 == b1) ? nA_up * (nA_up - 1.0) / 2.0 : nA_up * nB_up;
                            const double mean_up = ((double)kfa[k] * fmax) * del_t * np_up; // (the sweep's expression)
                            const double u_ = (double)w[k] * 0x1.0p-32;
                            if (ok && !(u_ < (1.0 - mean_up) - 4e-15)) pm |= 1u << k;
                        }
                    }
                    uint64_t anyp = __ballot(pm != 0);
                    while (anyp) {
                        const int l = __builtin_ctzll(anyp);
                        anyp &= anyp - 1;
                        const int cnt = __popc((uint32_t)__builtin_amdgcn_readlane((int)pm, l));
                        if (npre + cnt > cap_l - 1) { full = true; break; } // (the next round goes on behind the last entry)
                        if (lane == l) {
                            int o = npre;
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if ((pm >> k) & 1u)
                                    sl[o++] = make_uint2(((uint32_t)c1 << 24) | ((uint32_t)b1 << 16) | ((uint32_t)c2 << 8) | (uint32_t)(q4 + k), w[k]);
                        }
                        npre += cnt;
                    }
                }
            }
        }
        if (lane == 0) sl[npre] = make_uint2(full ? SL_FULL : SL_END, 0u);
    }
}

// ------------------------------------------------------------- species volumes
//
// aero_particle_coagulate's vol(:) = vol1(:) + vol2(:) [upstream, uncited; the container:
// interface/wrf_pmc_driver.F90:1419-1466].  No sampler reads a species volume (pairs are evaluated
// from the cached totals), so the sums are made here, behind the sampler, from the slots it logged
// (DeviceState::mlog) -- in event order wherever that matters: a wavefront per cell, 64 events at
// a time, each given the level 1 + the highest level among the earlier events that wrote a slot it
// reads (a particle that coagulated twice) and no level below that of an earlier event that read the
// slot it writes, the events of a level summed together, lanes over
// (event, species): 160 contiguous bytes per particle at the reference's 20 species, every load of
// a level in flight before its first store.  Same sums in the same order as a sampler that adds the
// species at the event (which it did until round 4: a lane per cell walking 20 species through five
// dependent round trips was three quarters of the sampler's time at 20 species).
__global__ __launch_bounds__(256) void k_vol_merge(DeviceState st)
{
    __shared__ uint32_t s_ev[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = st.n_spec;
    for (int c = blockIdx.x * 4 + wave; c < st.n_cells; c += gridDim.x * 4) {
        if (!in_loop(st, c)) continue;
        const int n_ev = min(st.n_events[c], st.max_events);
        const uint64_t *lg = st.mlog + (int64_t)c * st.max_events;
        double *vol = st.vol + (int64_t)c * st.cap * S;
        for (int e0 = 0; e0 < n_ev; e0 += 64) {
            const int ne = min(64, n_ev - e0);
            uint32_t pn = 0xFFFFFFFFu, po = 0xFFFFFFFEu;
            if (lane < ne) {
                const uint64_t ev = lg[e0 + lane];
                pn = (uint32_t)(ev & 0xFFFFu); po = (uint32_t)((ev >> 16) & 0xFFFFu);
            }
            int level = 0, top = 0;
            for (int i = 0; i + 1 < ne; i++) { // (uniform) event i's level is final when the loop reaches it
                const uint32_t pi = (uint32_t)__builtin_amdgcn_readlane((int)pn, i);
                const uint32_t qi = (uint32_t)__builtin_amdgcn_readlane((int)po, i);
                const int li = __builtin_amdgcn_readlane(level, i);
                if (lane > i && (pi == pn || pi == po)) level = max(level, li + 1);
                // (event i reads a slot this one writes -- its other particle, kept by the weighted removal:
                // not before it; the same level will do, a level's loads go out in event order before its stores)
                if (lane > i && qi == pn) level = max(level, li);
            }
            top = level;
            for (int d = 32; d >= 1; d >>= 1) top = max(top, __shfl_xor(top, d, 64));
            for (int lv = 0; lv <= top; lv++) {
                // the events of this level, side by side in LDS: pn | po << 16
                const uint64_t m = __ballot(lane < ne && level == lv);
                if (lane < ne && level == lv) s_ev[wave][__popcll(m & ((1ull << lane) - 1ull))] = pn | (po << 16);
                const int items = (int)__popcll(m) * S;
                __threadfence_block(); // (one wavefront: the LDS words; the stores of the level before this one)
                for (int it0 = 0; it0 < items; it0 += 256) {
                    double a[4], b[4];
                    int64_t dst[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int it = it0 + u * 64 + lane;
                        dst[u] = -1;
                        if (it < items) {
                            const int k = it / S, sp = it - k * S;
                            const uint32_t w = s_ev[wave][k];
                            dst[u] = (int64_t)(w & 0xFFFFu) * S + sp;
                            a[u] = vol[dst[u]];
                            b[u] = vol[(int64_t)(w >> 16) * S + sp];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (dst[u] >= 0) vol[dst[u]] = a[u] + b[u];
                }
                __threadfence_block();
            }
        }
    }
}

// ---------------------------------------------------------------- components
//
// particle%component(:) (interface/wrf_pmc_driver.F90:1421-1437).  The samplers only log which
// slots coagulated (DeviceState::mlog, a word per event); the lists are merged here, a thread per
// cell in event order: the merged particle's list is its first candidate's followed by its
// second's [aero_particle_coagulate, upstream, uncited], cut at n_comp_max (sum_s n_orig is the
// untruncated length).  Runs behind the sampler and before anything closes the cell's holes: the
// log names slots as the sampler saw them.
__global__ __launch_bounds__(64) void k_comp_merge(DeviceState st)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= st.n_cells || !in_loop(st, c)) return;
    const int n_ev = st.n_events[c];
    const int K = st.n_comp_max;
    const int64_t cb = (int64_t)c * st.cap;
    const uint64_t *lg = st.mlog + (int64_t)c * st.max_events;
    for (int e = 0; e < n_ev; e++) {
        const uint64_t ev = lg[e];
        const int pn = (int)(ev & 0xFFFFu), po = (int)((ev >> 16) & 0xFFFFu);
        const bool pn1 = ((ev >> 32) & 1u) != 0;
        const int64_t first = cb + (pn1 ? pn : po), second = cb + (pn1 ? po : pn);
        const uint64_t sa = st.comp_src[first], sb = st.comp_src[second];
        const int na = comp_count(sa), nb = comp_count(sb);
        double t[8];
        for (int k = 0; k < K; k++)
            t[k] = k < na ? st.comp_time[first * K + k]
                          : (k - na < nb ? st.comp_time[second * K + (k - na)] : 0.0);
        uint64_t sm = na >= 8 ? sa : ((sa & ((1ull << (8 * na)) - 1ull)) | (sb << (8 * na)));
        if (K < 8) sm |= ~0ull << (8 * K);
        st.comp_src[cb + pn] = sm;
        for (int k = 0; k < K; k++) st.comp_time[(cb + pn) * K + k] = t[k];
    }
}

// the host's view of the lists: len [n] entries held, source [n][K] (0-based, -1 past the end),
// time [n][K]; particles of cell k at part_offset[k].. (as pmcx_upload_cells)
__global__ void k_comp_upload(DeviceState st, int cell_first, const int64_t *part_offset,
                              const int32_t *len, const int32_t *source, const double *time)
{
    const int k = blockIdx.y, c = cell_first + k, K = st.n_comp_max;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0, cb = (int64_t)c * st.cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int m = len[o0 + i];
        m = m < 0 ? 0 : (m > K ? K : m);
        uint64_t w = ~0ull;
        for (int q = 0; q < K; q++) {
            if (q < m) w = (w & ~(0xFFull << (8 * q))) | ((uint64_t)(source[(o0 + i) * K + q] & 0xFF) << (8 * q));
            st.comp_time[(cb + i) * K + q] = q < m ? time[(o0 + i) * K + q] : 0.0;
        }
        st.comp_src[cb + i] = w;
    }
}
__global__ void k_comp_download(DeviceState st, int cell_first, const int64_t *part_offset,
                                int32_t *len, int32_t *source, double *time)
{
    const int k = blockIdx.y, c = cell_first + k, K = st.n_comp_max;
    const int64_t o0 = part_offset[k], n = part_offset[k + 1] - o0, cb = (int64_t)c * st.cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t w = st.comp_src[cb + i];
        len[o0 + i] = comp_count(w);
        for (int q = 0; q < K; q++) {
            const uint32_t b = (uint32_t)(w >> (8 * q)) & 0xFFu;
            source[(o0 + i) * K + q] = b == 0xFFu ? -1 : (int32_t)b;
            time[(o0 + i) * K + q] = st.comp_time[(cb + i) * K + q];
        }
    }
}
// what an upload without lists leaves: n_orig(s) entries per source in source order, the first
// created at the least create time and the others at the greatest (exact for a particle of
// one constituent)
__global__ void k_comp_default(DeviceState st, int cell_first)
{
    const int c = cell_first + blockIdx.y, K = st.n_comp_max;
    const int64_t cb = (int64_t)c * st.cap;
    const int n = st.n_hi[c];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t w = ~0ull;
        int m = 0;
        for (int s = 0; s < st.n_source && m < K; s++)
            for (int q = norig(st, cb, s, i); q > 0 && m < K; q--, m++)
                w = (w & ~(0xFFull << (8 * m))) | ((uint64_t)s << (8 * m));
        for (int q = 0; q < K; q++)
            st.comp_time[(cb + i) * K + q] = q >= m ? 0.0 : (q == 0 ? st.hot[cb + i].lct : st.hot[cb + i].gct);
        st.comp_src[cb + i] = w;
    }
}

// ---------------------------------------------------------------- compaction
//
// Removed particles become tombstones (cached volume -1): slots [0, n_hi) hold
// n_part live particles in a fixed relative order, and holes are closed by a
// stable compaction only when more than 1/compact_den of the slots are dead (or when a
// caller needs a dense cell).  Because the order of the survivors never
// changes, when compaction happens cannot change any later result.

__device__ __forceinline__ void move_particle(const DeviceState &st, int64_t cb, int c, int dst,
                                              int src)
{
    // every record is read before any is written: the loads go out together (written field by
    // field, each store would have to wait for the load behind it -- source and destination are
    // the same arrays to the compiler)
    const Hot h = st.hot[cb + src];
    const Cold cd = st.cold[cb + src];
    const uint16_t gk = st.gkey[cb + src];
    const int32_t wc = st.wclass[cb + src];
    double v4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < 4 && s < st.n_spec; s++) v4[s] = st.vol[(cb + src) * st.n_spec + s];
    int32_t no4[4] = {0, 0, 0, 0};
    for (int s = 1; s < 4 && s < st.n_source; s++) no4[s] = norig(st, cb, s, src);
    st.hot[cb + dst] = h;
    st.cold[cb + dst] = cd;
    st.gkey[cb + dst] = gk;
    st.wclass[cb + dst] = wc;
    for (int s = 0; s < 4 && s < st.n_spec; s++) st.vol[(cb + dst) * st.n_spec + s] = v4[s];
    for (int s = 1; s < 4 && s < st.n_source; s++) norig(st, cb, s, dst) = no4[s];
    for (int s = 4; s < st.n_spec; s++)
        st.vol[(cb + dst) * st.n_spec + s] = st.vol[(cb + src) * st.n_spec + s];
    for (int s = 4; s < st.n_source; s++)
        norig(st, cb, s, dst) = norig(st, cb, s, src);
    comp_copy(st, cb + dst, cb + src);
}

// Stable compaction of one cell by one wavefront: 64 slots at a time, every
// row read into registers before any is written (dst <= src, chunks ascend).
// The slots are renamed by a monotone map; when the caller passes a chunk
// table (LDS, n_hi/64 + 1 entries each) and the cell's lists hold, their
// entries follow the particles, so that closing holes is invisible to the
// sampler.  Callers that change the population pass nullptr and clear
// sort_valid themselves.
__device__ void compact_wave(const DeviceState &st, int c, int lane, uint32_t *cbase = nullptr,
                             uint64_t *cmask = nullptr, bool map_always = false)
{
    const int64_t cb = (int64_t)c * st.cap;
    const int n_hi = st.n_hi[c];
    const int S = st.n_spec, NS = st.n_source;
    const uint64_t lt = (1ull << lane) - 1ull;
    apply_event_log(st, c, lane); // (its entries name slots as they are numbered now)
    // the slot map (chunk bases and live masks) is filled for the caller when the cell's lists
    // hold (they are renamed below) or when the caller renames something itself (map_always)
    const bool lists = cbase != nullptr && st.sort_valid[c] != 0;
    const bool remap = lists || (cbase != nullptr && map_always);
    int base = 0;
    // Four 64-slot chunks per trip.  A trip's sources are slots nobody has written (destinations
    // lie at or below their sources, and below the next trip's first slot): the only order that
    // matters is inside a trip -- every field of its movers is in registers before the first is
    // stored.  So a trip waits for memory ONCE: the liveness words of the NEXT trip are requested
    // before this trip's fields, the source counts (four sources at a time) with them, and the
    // stores drain while the next trip works.
    constexpr int U = 4;
    double xn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int i = u * 64 + lane;
        xn[u] = i < n_hi ? ld_f64(&st.hot[cb + i].vtot) : -1.0;
    }
    for (int i0 = 0; i0 < n_hi; i0 += 64 * U) {
        double x[U];
        int dst[U];
        bool mv[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = xn[u];
#pragma unroll
        for (int u = 0; u < U; u++) { // (the next trip's liveness: on its way while this trip moves)
            const int i = i0 + 64 * U + u * 64 + lane;
            xn[u] = i < n_hi ? ld_f64(&st.hot[cb + i].vtot) : -1.0;
        }
        bool any = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * 64 + lane;
            const bool live = x[u] >= 0.0; // tombstones carry a negative cached volume
            const uint64_t m = __ballot(live);
            dst[u] = base + __popcll(m & lt);
            mv[u] = live && dst[u] != i;
            if (remap && lane == 0 && i0 + u * 64 < n_hi) {
                cbase[(i0 >> 6) + u] = (uint32_t)base;
                cmask[(i0 >> 6) + u] = m;
            }
            base += __popcll(m);
            any |= __ballot(mv[u]) != 0;
        }
        if (!any) continue;
        double a1[U], a2[U], a3[U];
        int64_t idv[U];
        int32_t wc[U];
        uint16_t kk[U];
        uint32_t xo[4][U]; // the first four sources' counts travel with the records
        const int NS4 = NS < 4 ? NS : 4;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * 64 + lane;
            a1[u] = a2[u] = a3[u] = 0; idv[u] = 0; wc[u] = 0; kk[u] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) xo[q][u] = 0u;
            if (mv[u]) {
                kk[u] = st.gkey[cb + i];
                a1[u] = ld_f64(&st.hot[cb + i].mass);
                a2[u] = ld_f64(&st.hot[cb + i].lct); a3[u] = ld_f64(&st.hot[cb + i].gct);
                idv[u] = (int64_t)d2u(ld_f64((const double *)&st.cold[cb + i].id));
                wc[u] = (int32_t)ld_u32((const uint32_t *)&st.wclass[cb + i]);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < NS4) xo[q][u] = ld_u32((const uint32_t *)&norig(st, cb, q, i));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
#pragma unroll
        for (int u = 0; u < U; u++)
            if (mv[u]) {
                const int d = dst[u];
                st_f64(&st.hot[cb + d].vtot, x[u]); st_f64(&st.hot[cb + d].mass, a1[u]);
                st_f64(&st.hot[cb + d].lct, a2[u]); st_f64(&st.hot[cb + d].gct, a3[u]);
                st_f64((double *)&st.cold[cb + d].id, u2d((uint64_t)idv[u]));
                st_u32((uint32_t *)&st.wclass[cb + d], (uint32_t)wc[u]);
                st.gkey[cb + d] = kk[u];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < NS4) st_u32((uint32_t *)&norig(st, cb, q, d), xo[q][u]);
            }
        // species volumes (with one species the plane is not kept current: nothing to move)
        if (S > 1)
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * 64 + lane;
                if (__ballot(mv[u]) == 0) continue;
                for (int s0 = 0; s0 < S; s0 += 4) {
                    double v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        v[k] = (mv[u] && s0 + k < S) ? ld_f64(&st.vol[(cb + i) * S + (s0 + k)]) : 0.0;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    wave_sync();
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (mv[u] && s0 + k < S) st_f64(&st.vol[(cb + dst[u]) * S + (s0 + k)], v[k]);
                }
            }
        for (int s = 4; s < NS; s++) { // (sources beyond the fourth: one at a time)
            uint32_t xs[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                xs[u] = mv[u] ? ld_u32((const uint32_t *)&norig(st, cb, s, i0 + u * 64 + lane)) : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_sync();
#pragma unroll
            for (int u = 0; u < U; u++)
                if (mv[u]) st_u32((uint32_t *)&norig(st, cb, s, dst[u]), xs[u]);
        }
        if (st.n_comp_max) { // the component lists: the source word, then one create time at a time
            const int K = st.n_comp_max;
            for (int k = -1; k < K; k++) {
                double xc[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int64_t at = cb + i0 + u * 64 + lane;
                    xc[u] = !mv[u] ? 0.0 : k < 0 ? ld_f64((const double *)&st.comp_src[at])
                                                  : ld_f64(&st.comp_time[at * K + k]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (mv[u]) {
                        if (k < 0) st_f64((double *)&st.comp_src[cb + dst[u]], xc[u]);
                        else st_f64(&st.comp_time[(cb + dst[u]) * K + k], xc[u]);
                    }
            }
        }
    }
    if (lane == 0) st.n_hi[c] = base;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lists && base != n_hi) {
        // every position of perm up to the end of the last list; room behind a list holds
        // stale entries, renaming them is harmless
        uint16_t *perm = st.perm + (int64_t)c * st.pcap;
        const int G = st.n_bin * st.n_class;
        const int used = st.goff[(int64_t)c * (G + 1) + G];
        for (int j = lane; j < used; j += 64) {
            const uint32_t e = perm[j];
            if (e < (uint32_t)n_hi) {
                const uint32_t k = e >> 6;
                perm[j] = (uint16_t)(cbase[k] + (uint32_t)__popcll(cmask[k] & ((1ull << (e & 63)) - 1ull)));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
    }
}

// The same compaction by a whole workgroup of GW wavefronts (256 threads) on ONE cell: a trip is
// GW x GU chunks of 64 slots, chunk q = u GW + wave of it belonging to `wave`, the running base
// handed round through LDS.  One wavefront's compaction of a 10 000-slot cell is 39 dependent
// trips; when a few cells of a block need one (a cell thinned by the preweight, one that makes
// room for arrivals), those chains are the kernel's whole duration -- here they are a quarter as
// long.  Same slot map, same result.  s_cnt: GW * GU + 2 ints of LDS.  Every thread of the
// workgroup calls it with the same c (barriers inside).
constexpr int GW = 4, GU = 2;
constexpr int COMPACT_GROUP_MAX = 4096; // listed cells up to which k_compact takes a workgroup per cell
__device__ void compact_group(const DeviceState &st, int c, int tid, uint32_t *cbase, uint64_t *cmask,
                              bool map_always, int *s_cnt)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t cb = (int64_t)c * st.cap;
    const int S = st.n_spec, NS = st.n_source;
    const uint64_t lt = (1ull << lane) - 1ull;
    if (wave == 0) apply_event_log(st, c, lane);
    __syncthreads();
    const int n_hi = st.n_hi[c];
    const bool lists = st.sort_valid[c] != 0;
    const bool remap = lists || map_always;
    int base = 0;
    constexpr int T = 64 * GW * GU; // slots per trip
    double xn[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
        const int i = (u * GW + wave) * 64 + lane;
        xn[u] = i < n_hi ? ld_f64(&st.hot[cb + i].vtot) : -1.0;
    }
    for (int i0 = 0; i0 < n_hi; i0 += T) {
        double x[GU];
        int dst[GU], cnt[GU];
        bool mv[GU];
        uint64_t m[GU];
#pragma unroll
        for (int u = 0; u < GU; u++) x[u] = xn[u];
#pragma unroll
        for (int u = 0; u < GU; u++) { // (the next trip's liveness: on its way while this trip moves)
            const int i = i0 + T + (u * GW + wave) * 64 + lane;
            xn[u] = i < n_hi ? ld_f64(&st.hot[cb + i].vtot) : -1.0;
        }
#pragma unroll
        for (int u = 0; u < GU; u++) {
            m[u] = __ballot(x[u] >= 0.0); // tombstones carry a negative cached volume
            cnt[u] = __popcll(m[u]);
            if (lane == 0) s_cnt[u * GW + wave] = cnt[u];
        }
        __syncthreads();
        int before[GU] = {}, total = 0;
#pragma unroll
        for (int q = 0; q < GW * GU; q++) {
            const int v = s_cnt[q];
#pragma unroll
            for (int u = 0; u < GU; u++)
                if (q == u * GW + wave) before[u] = total;
            total += v;
        }
#pragma unroll
        for (int u = 0; u < GU; u++) {
            const int q = u * GW + wave, i = i0 + q * 64 + lane;
            dst[u] = base + before[u] + __popcll(m[u] & lt);
            mv[u] = x[u] >= 0.0 && dst[u] != i;
            if (remap && lane == 0 && i0 + q * 64 < n_hi) {
                cbase[(i0 >> 6) + q] = (uint32_t)(base + before[u]);
                cmask[(i0 >> 6) + q] = m[u];
            }
        }
        // nothing can move while no slot up to the end of this trip is dead (known to every
        // wavefront alike: no word passes between them for it)
        const bool dense = base == i0 && total == (n_hi - i0 < T ? n_hi - i0 : T);
        base += total;
        if (dense) {
            __syncthreads(); // (s_cnt is rewritten by the next trip)
            continue;
        }
        double a1[GU], a2[GU], a3[GU];
        int64_t idv[GU];
        int32_t wc[GU];
        uint16_t kk[GU];
        uint32_t xo[4][GU];
        const int NS4 = NS < 4 ? NS : 4;
#pragma unroll
        for (int u = 0; u < GU; u++) {
            const int i = i0 + (u * GW + wave) * 64 + lane;
            a1[u] = a2[u] = a3[u] = 0; idv[u] = 0; wc[u] = 0; kk[u] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) xo[q][u] = 0u;
            if (mv[u]) {
                kk[u] = st.gkey[cb + i];
                a1[u] = ld_f64(&st.hot[cb + i].mass);
                a2[u] = ld_f64(&st.hot[cb + i].lct); a3[u] = ld_f64(&st.hot[cb + i].gct);
                idv[u] = (int64_t)d2u(ld_f64((const double *)&st.cold[cb + i].id));
                wc[u] = (int32_t)ld_u32((const uint32_t *)&st.wclass[cb + i]);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < NS4) xo[q][u] = ld_u32((const uint32_t *)&norig(st, cb, q, i));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // every wavefront holds its movers: their slots may be written
#pragma unroll
        for (int u = 0; u < GU; u++)
            if (mv[u]) {
                const int d = dst[u];
                st_f64(&st.hot[cb + d].vtot, x[u]); st_f64(&st.hot[cb + d].mass, a1[u]);
                st_f64(&st.hot[cb + d].lct, a2[u]); st_f64(&st.hot[cb + d].gct, a3[u]);
                st_f64((double *)&st.cold[cb + d].id, u2d((uint64_t)idv[u]));
                st_u32((uint32_t *)&st.wclass[cb + d], (uint32_t)wc[u]);
                st.gkey[cb + d] = kk[u];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < NS4) st_u32((uint32_t *)&norig(st, cb, q, d), xo[q][u]);
            }
        if (S > 1)
            for (int s0 = 0; s0 < S; s0 += 4) {
                double v[GU][4];
#pragma unroll
                for (int u = 0; u < GU; u++)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        v[u][k] = (mv[u] && s0 + k < S)
                                      ? ld_f64(&st.vol[(cb + i0 + (u * GW + wave) * 64 + lane) * S + (s0 + k)]) : 0.0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int u = 0; u < GU; u++)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (mv[u] && s0 + k < S) st_f64(&st.vol[(cb + dst[u]) * S + (s0 + k)], v[u][k]);
            }
        for (int sq = 4; sq < NS; sq++) { // (sources beyond the fourth: one at a time)
            uint32_t xs[GU];
#pragma unroll
            for (int u = 0; u < GU; u++)
                xs[u] = mv[u] ? ld_u32((const uint32_t *)&norig(st, cb, sq, i0 + (u * GW + wave) * 64 + lane)) : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int u = 0; u < GU; u++)
                if (mv[u]) st_u32((uint32_t *)&norig(st, cb, sq, dst[u]), xs[u]);
        }
        if (st.n_comp_max) { // the component lists: the source word, then one create time at a time
            const int K = st.n_comp_max;
            for (int k = -1; k < K; k++) {
                double xc[GU];
#pragma unroll
                for (int u = 0; u < GU; u++) {
                    const int64_t at = cb + i0 + (u * GW + wave) * 64 + lane;
                    xc[u] = !mv[u] ? 0.0 : k < 0 ? ld_f64((const double *)&st.comp_src[at])
                                                  : ld_f64(&st.comp_time[at * K + k]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int u = 0; u < GU; u++)
                    if (mv[u]) {
                        if (k < 0) st_f64((double *)&st.comp_src[cb + dst[u]], xc[u]);
                        else st_f64(&st.comp_time[(cb + dst[u]) * K + k], xc[u]);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st.n_hi[c] = base;
    if (lists && base != n_hi) {
        uint16_t *perm = st.perm + (int64_t)c * st.pcap;
        const int G = st.n_bin * st.n_class;
        const int used = st.goff[(int64_t)c * (G + 1) + G];
        for (int j = tid; j < used; j += 64 * GW) {
            const uint32_t e = perm[j];
            if (e < (uint32_t)n_hi) {
                const uint32_t k = e >> 6;
                perm[j] = (uint16_t)(cbase[k] + (uint32_t)__popcll(cmask[k] & ((1ull << (e & 63)) - 1ull)));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// compacts the cells whose dead fraction exceeds num/den (num = 0: any dead slot)
__global__ __launch_bounds__(256) void k_compact(DeviceState st, const int32_t *list,
                                                 const int32_t *count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_chunk = (st.cap + 63) / 64 + 1;
    uint64_t *cmask = (uint64_t *)smem + (size_t)wave * n_chunk;
    uint32_t *cbase = (uint32_t *)((uint64_t *)smem + (size_t)(blockDim.x >> 6) * n_chunk) +
                      (size_t)wave * n_chunk;
    const int n_work = *count, stride = gridDim.x * (blockDim.x >> 6);
    // a few cells: a workgroup each (the duration is one cell's chain of trips, a quarter as long
    // that way); many: a wavefront each (the same work with no barriers)
    if (n_work <= COMPACT_GROUP_MAX) {
        uint64_t *cmask0 = (uint64_t *)smem;
        uint32_t *cbase0 = (uint32_t *)(cmask0 + (size_t)GW * n_chunk);
        int *s_cnt = (int *)(cbase0 + (size_t)GW * n_chunk);
        for (int w = blockIdx.x; w < n_work; w += gridDim.x)
            compact_group(st, list[w], threadIdx.x, cbase0, cmask0, false, s_cnt);
        return;
    }
    for (int w = blockIdx.x * (blockDim.x >> 6) + wave; w < n_work; w += stride) {
        compact_wave(st, list[w], lane, cbase, cmask);
        wave_sync();
    }
}

// ---------------------------------------------------------------- rebalance
//
// aero_state_rebalance (interface/wrf_pmc_driver.F90:248-250): per class,
// double (copy every particle, new ids, halve the weight) while the class
// holds fewer than n_ideal/2 particles, halve (drop each with probability
// 1/2, double the weight) while it holds more than 2 n_ideal.  Dropped
// particles become tombstones; the halving draw of a particle is keyed by its
// rank among the live particles, which does not depend on where holes are.

__global__ __launch_bounds__(64) void k_rebalance(DeviceState st, int allow_doubling,
                                                  int allow_halving, uint64_t seed, uint32_t step,
                                                  const int32_t *list, const int32_t *count)
{
    const int lane = threadIdx.x;
    const int n_work = list ? *count : st.n_cells;
  for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const int c = list ? list[wi] : wi;
    if (!in_loop(st, c)) continue; // (uniform over the wavefront)
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << lane) - 1ull;
    double *mag = st.mag + (int64_t)c * st.n_class;
    const double *n_ideal = st.n_ideal + (int64_t)c * st.n_class;
    int status = 0;
    apply_event_log(st, c, lane);
    // particles per class, class q in lane q (up to 64 classes): the lengths of the cell's lists while they
    // hold (what mc_coag leaves), else ONE pass with a ballot per class -- not a pass per class and
    // doubling, which with ten classes was 247 ms of a 535-ms step at the full domain
    const bool counted = st.n_class > 1;
    __shared__ int s_cls[8 * 256];
    if (counted) wave_class_counts(st, c, lane, s_cls);
    for (int cl = 0; cl < st.n_class; cl++) {
        int ncl_kept = counted ? s_cls[cl] : 0;
        for (uint32_t iter = 0; iter < 32; iter++) {
            int n_hi = st.n_hi[c];
            const int n_live = st.n_part[c];
            int ncl = n_live; // one class: every live particle belongs to it
            if (counted) {
                ncl = ncl_kept;
            } else if (st.n_class > 1) {
                uint32_t cnt = 0;
                for (int i = lane; i < n_hi; i += 64) cnt += (live_class_c(st, cb, i) == cl);
                ncl = (int)wave_sum_u64(cnt);
            }
            const double ideal = n_ideal[cl];
            if (allow_doubling && ncl > 0 && (double)ncl < ideal / 2.0) {
                if (n_hi + ncl > st.cap && n_hi > n_live) { // make room: close the holes first
                    compact_wave(st, c, lane);
                    n_hi = n_live;
                }
                if (n_hi + ncl > st.cap) { status |= ST_CAP_OVERFLOW; break; }
                const int64_t nid = st.next_id[c];
                uint32_t base = 0;
                int wc_next = lane < n_hi ? live_class_c(st, cb, lane) : -1; // (requested a chunk ahead)
                for (int i0 = 0; i0 < n_hi; i0 += 64) {
                    const int i = i0 + lane;
                    const bool is = wc_next == cl;
                    wc_next = i + 64 < n_hi ? live_class_c(st, cb, i + 64) : -1;
                    const uint64_t m = __ballot(is);
                    const uint32_t k = base + (uint32_t)__popcll(m & lt);
                    if (is) {
                        move_particle(st, cb, c, n_hi + (int)k, i);
                        st.cold[cb + n_hi + k].id = nid + (int64_t)k;
                    }
                    base += (uint32_t)__popcll(m);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                if (lane == 0) {
                    st.next_id[c] = nid + ncl;
                    st.n_hi[c] = n_hi + ncl;
                    st.n_part[c] = n_live + ncl;
                    mag[cl] = mag[cl] / 2.0;
                    st.sort_valid[c] = 0;
                }
                ncl_kept = 2 * ncl;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __threadfence_block();
                wave_sync();
            } else if (allow_halving && (double)ncl > ideal * 2.0) {
                uint32_t nd = 0, rank0 = 0;
                int wc_next = lane < n_hi ? live_class_c(st, cb, lane) : -1; // (requested a chunk ahead)
                for (int i0 = 0; i0 < n_hi; i0 += 64) {
                    const int i = i0 + lane;
                    const int wc = wc_next;
                    wc_next = i + 64 < n_hi ? live_class_c(st, cb, i + 64) : -1;
                    const uint64_t mlive = __ballot(wc >= 0);
                    const uint32_t rank = rank0 + (uint32_t)__popcll(mlive & lt);
                    bool drop = false;
                    if (wc == cl) {
                        uint32_t w[4];
                        rng_block(seed, cell_id, step, rank, SITE_HALVE, (uint32_t)cl * 32u + iter, w);
                        drop = u53(w[0], w[1]) < 0.5;
                    }
                    if (drop) mark_dead(st, cb, i);
                    nd += (uint32_t)__popcll(__ballot(drop));
                    rank0 += (uint32_t)__popcll(mlive);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                if (lane == 0) {
                    st.n_part[c] = n_live - (int)nd;
                    mag[cl] = mag[cl] * 2.0;
                    st.sort_valid[c] = 0;
                }
                ncl_kept = ncl - (int)nd;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __threadfence_block();
                wave_sync();
                // (past the tombstone threshold: the k_compact that follows)
            } else {
                break;
            }
        }
    }
    if (lane == 0 && status) atomicOr(&st.status[c], status);
    wave_sync();
  }
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
    const int c = blockIdx.x, t = threadIdx.x, wave = t >> 6;
    const int nd = st.n_dbin, C = st.n_class, S = st.n_spec;
    // histograms are private to each of the 4 wavefronts (fewer atomic collisions)
    double *edge = (double *)smem;                       // [nd+1]
    double *msum = edge + (nd + 1);                      // [4][nd]
    double *red = msum + 4 * nd;                         // [256]
    uint32_t *ccnt = (uint32_t *)(red + 256);            // [4][C][nd]
    uint32_t *clcnt = ccnt + 4 * (size_t)C * nd;         // [C]
    const size_t tile_off = ((size_t)((unsigned char *)(clcnt + C) - smem) + 15) & ~(size_t)15;
    for (int j = t; j <= nd; j += 256) edge[j] = st.d_edge[j];
    for (int j = t; j < 4 * nd; j += 256) msum[j] = 0.0;
    for (int j = t; j < 4 * C * nd; j += 256) ccnt[j] = 0;
    for (int j = t; j < C; j += 256) clcnt[j] = 0;
    __syncthreads();
    const int n = st.n_hi[c];
    const int64_t cb = (int64_t)c * st.cap;
    const double *mag = st.mag + (int64_t)c * C;
    double *my_msum = msum + wave * nd;
    uint32_t *my_ccnt = ccnt + (size_t)wave * C * nd;
    double tm = 0.0;
    uint32_t n_live = 0;
    double *tile = (double *)(smem + tile_off) + (size_t)wave * 64 * (S + 1);
    const int lane = t & 63;
    for (int i0 = t & ~63; i0 < n; i0 += 256) {
        const int i = i0 + lane;
        if (S > 1) stage_species(st, cb, i0, n, tile, lane);
        if (i >= n) continue;
        const int cl = live_class(st, cb, i);
        if (cl < 0) continue; // tombstone
        double vdry = 0.0, ms = 0.0;
        const double v_only = st.hot[cb + i].vtot; // (one species: its volume is the cached total)
        for (int s = 0; s < S; s++) {
            const double v = S > 1 ? tile[lane * (S + 1) + s] : v_only;
            if (s != st.i_water) vdry = vdry + v;
            ms = ms + st.density[s] * v;
        }
        const double nc = mag[cl];
        const double diam = 2.0 * det_cbrt(vdry * (0.75 / PI));
        if (C == 1) n_live++; else atomicAdd(&clcnt[cl], 1u);
        tm += ms * nc;
        if (diam >= edge[0] && diam < edge[nd]) {
            int b;
            if (st.dbin_fast) { // float estimate of log2(diam), settled against the two edges
                const uint32_t hi = (uint32_t)(d2u(diam) >> 32);
                const float l2 = (float)((int)((hi >> 20) & 0x7FF) - 1023) +
                                 (float)(hi & 0xFFFFFu) * 0x1.0p-20f;
                b = (int)((l2 - st.dlog2_d0) * st.dinv_dlog2);
                b = b < 0 ? 0 : (b > nd - 1 ? nd - 1 : b);
                b += (int)(diam >= edge[b + 1]) - (int)(diam < edge[b]);
                b = b < 0 ? 0 : (b > nd - 1 ? nd - 1 : b);
            } else {
                b = bin_of(edge, nd, diam);
            }
            atomicAdd(&my_ccnt[cl * nd + b], 1u);
            atomicAdd(&my_msum[b], nc * ms);
        }
    }
    if (C == 1 && n_live) atomicAdd(&clcnt[0], n_live);
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
        for (int cl = 0; cl < C; cl++) {
            const uint32_t k = ccnt[cl * nd + b] + ccnt[(C + cl) * nd + b] +
                               ccnt[(2 * C + cl) * nd + b] + ccnt[(3 * C + cl) * nd + b];
            x = x + (double)k * mag[cl];
        }
        num_dist[(int64_t)c * nd + b] = x / st.d_width[b];
        mass_dist[(int64_t)c * nd + b] =
            (((msum[b] + msum[nd + b]) + msum[2 * nd + b]) + msum[3 * nd + b]) / st.d_width[b];
    }
}

// extended observables: species mass concentrations (driver:864-932), number
// concentration by source (driver:858-861), PM1/PM2.5/PM10 (driver:1153-1157)
__global__ __launch_bounds__(256) void k_diagnostics_ext(DeviceState st, double *species_mass,
                                                         double *source_num, double *pm)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int c = blockIdx.x, t = threadIdx.x;
    const int S = st.n_spec, NS = st.n_source;
    double *acc = (double *)smem; // [S] species, [NS] sources, [3] pm
    for (int j = t; j < S + NS + 3; j += 256) acc[j] = 0.0;
    __syncthreads();
    const int n = st.n_hi[c];
    const int64_t cb = (int64_t)c * st.cap;
    const double *mag = st.mag + (int64_t)c * st.n_class;
    double *tile = (double *)(smem + (((size_t)(S + NS + 3) * sizeof(double) + 15) & ~(size_t)15)) +
                   (size_t)(t >> 6) * 64 * (S + 1);
    const int lane = t & 63;
    for (int i0 = t & ~63; i0 < n; i0 += 256) {
        const int i = i0 + lane;
        if (S > 1) stage_species(st, cb, i0, n, tile, lane);
        if (i >= n) continue;
        const int cl = live_class(st, cb, i);
        if (cl < 0) continue;
        const double nc = mag[cl];
        double vdry = 0.0, mdry = 0.0;
        const double v_only = st.hot[cb + i].vtot; // (one species: its volume is the cached total)
        for (int s = 0; s < S; s++) {
            const double v = S > 1 ? tile[lane * (S + 1) + s] : v_only;
            atomicAdd(&acc[s], st.density[s] * v * nc);
            if (s != st.i_water) { vdry = vdry + v; mdry = mdry + st.density[s] * v; }
        }
        for (int k = 0; k < NS; k++)
            if (norig(st, cb, k, i) > 0) atomicAdd(&acc[S + k], nc);
        const double diam = 2.0 * det_cbrt(vdry * (0.75 / PI));
        if (diam < 1e-6) atomicAdd(&acc[S + NS], nc * mdry);
        if (diam < 2.5e-6) atomicAdd(&acc[S + NS + 1], nc * mdry);
        if (diam < 10e-6) atomicAdd(&acc[S + NS + 2], nc * mdry);
    }
    __syncthreads();
    for (int j = t; j < S; j += 256) species_mass[(int64_t)c * S + j] = acc[j];
    for (int j = t; j < NS; j += 256) source_num[(int64_t)c * NS + j] = acc[S + j];
    for (int j = t; j < 3; j += 256) pm[(int64_t)c * 3 + j] = acc[S + NS + j];
}

// ----------------------------------------------------------------- scenario
//
// scenario_update_aero_state (interface/wrf_pmc_driver.F90:207-210)
// [upstream: aero_state_add_aero_dist_sample, aero_state_sample_particles]:
// emission of new particles from log-normal modes, loss by dilution, gain
// from the background.  One wavefront per cell; new particles are appended in
// order at n_hi, lost ones become tombstones.

struct AeroDistDev {
    int32_t n_mode;
    const double *num_conc;     // [n_cells][n_mode]
    const double *ln_gm_radius, *ln_gsd, *vol_frac;
    const int32_t *mode_class, *mode_source;
    // 0: the mode parameters are [n_mode] ([n_mode][n_spec]), shared by every cell; 1: every cell has
    // its own, [n_cells][n_mode] (the reference reads a cell's emission and boundary modes from that
    // cell's file, wrf_pmc_init.F90:583-804, 809-967)
    int32_t per_cell;
    __device__ __forceinline__ int64_t mo(int c) const { return per_cell ? (int64_t)c * n_mode : 0; }
};

// aero_state_scale_weight(weight_ratio) of one class by one wavefront [upstream, uncited]:
//   ratio > 1 (and allow_halving, or the class is empty): exactly prob_round(n (1 - 1 / ratio)) of the
//     class's n particles leave, a uniformly random subset (selection sampling in rank order:
//     member t goes when u (n - t) < still to go); they become tombstones;
//   ratio < 1 (and allow_doubling, or the class is empty): every particle of the class stands for
//     prob_round(1 / ratio) >= 1: its copies, under new ids, are appended in rank order;
//   then the magnitude is multiplied by ratio.  bkey separates the callers' random numbers (the
//   transport's preweight: the class; an add: SCALE_ADD | kind << 16 | mode << 8 | class).
// Returns the number removed or added.  The caller has applied the event log.
constexpr uint32_t SCALE_ADD = 0x4000000u;
// COPIES = false: the caller only ever thins (the transport's preweight) and is spared the other
// branch's code and registers.
template <bool COPIES>
__device__ int scale_weight_wave(const DeviceState &st, int c, int cl, double ratio, int allow_doubling,
                                 int allow_halving, uint64_t seed, uint32_t step, uint32_t bkey, int lane,
                                 int &status)
{
    const int C = st.n_class;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << lane) - 1ull;
    double *mag = st.mag + (int64_t)c * C;
    int n_hi = st.n_hi[c];
    const int n_live = st.n_part[c];
    int n_before = n_live;
    if (C > 1) {
        uint32_t cnt = 0;
        for (int i = lane; i < n_hi; i += 64) cnt += (live_class_c(st, cb, i) == cl);
        n_before = (int)wave_sum_u64(cnt);
    }
    uint32_t w[4];
    int changed = 0;
    if (ratio > 1.0 && (allow_halving || n_before == 0)) {
        rng_block(seed, cell_id, step, 0xFFFFFFFFu, SITE_SCALE, bkey, w);
        const double x_rm = (double)n_before * (1.0 - 1.0 / ratio);
        const double x_fl = (double)(int64_t)x_rm;
        int m_rm = (int)x_fl + ((u53(w[0], w[1]) < x_rm - x_fl) ? 1 : 0);
        m_rm = m_rm < n_before ? m_rm : n_before;
        uint32_t rank0 = 0;
        int t0 = 0, k_rm = 0; // class members seen, removed so far (uniform)
        // (the next chunk's liveness is requested before this chunk's work: a wavefront alone with a
        // big cell would otherwise wait out a memory round trip per 64 slots)
        int wc_next = lane < n_hi ? live_class_c(st, cb, lane) : -1;
        for (int i0 = 0; i0 < n_hi && k_rm < m_rm; i0 += 64) {
            const int i = i0 + lane;
            const int wc = wc_next;
            wc_next = i + 64 < n_hi ? live_class_c(st, cb, i + 64) : -1;
            const uint64_t mlive = __ballot(wc >= 0);
            const uint64_t mcls = __ballot(wc == cl);
            double u = 0.0;
            if (wc == cl) {
                rng_block(seed, cell_id, step, rank0 + (uint32_t)__popcll(mlive & lt), SITE_SCALE, bkey, w);
                u = u53(w[0], w[1]);
            }
            // (member t goes when u (n - t) < m - k, i.e. when k <= m - 1 - floor(u (n - t)): a bound every
            // lane works out for itself; the serial part is one scalar compare a member)
            const int t_l = t0 + __popcll(mcls & lt);
            const int a_l = (wc == cl) ? m_rm - 1 - (int)(u * (double)(n_before - t_l)) : -1;
            uint64_t msel = 0, rest = mcls;
            while (rest) {
                const int l = __builtin_ctzll(rest);
                rest &= rest - 1;
                if (k_rm <= __builtin_amdgcn_readlane(a_l, l)) { msel |= 1ull << l; k_rm++; }
            }
            t0 += __popcll(mcls);
            if ((msel >> lane) & 1ull) mark_dead(st, cb, i);
            rank0 += (uint32_t)__popcll(mlive);
        }
        changed = k_rm;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0) st.n_part[c] = n_live - changed;
    } else if (COPIES && ratio < 1.0 && (allow_doubling || n_before == 0)) {
        // (no particle stands for more than the cell has slots: the count stays an integer whatever
        // the ratio, and a cell that cannot hold the copies is flagged full either way)
        const double x = fmin(1.0 / ratio, (double)st.cap), x_fl = (double)(int64_t)x;
        // how many copies a class member gets (its uniform is keyed by its rank: the same in both passes)
        auto dups_of = [&](int wc, uint32_t rank) -> uint32_t {
            if (wc != cl) return 0u;
            uint32_t q[4];
            rng_block(seed, cell_id, step, rank, SITE_SCALE, bkey, q);
            return (uint32_t)((int)x_fl + ((u53(q[0], q[1]) < x - x_fl) ? 1 : 0) - 1);
        };
        uint64_t total = 0; // (up to cap copies of up to cap particles)
        uint32_t rank0 = 0;
        for (int i0 = 0; i0 < n_hi; i0 += 64) {
            const int i = i0 + lane;
            const int wc = i < n_hi ? live_class_c(st, cb, i) : -1;
            const uint64_t mlive = __ballot(wc >= 0);
            total += wave_sum_u64(dups_of(wc, rank0 + (uint32_t)__popcll(mlive & lt)));
            rank0 += (uint32_t)__popcll(mlive);
        }
        if ((uint64_t)n_hi + total > (uint64_t)st.cap && n_hi > n_live) { // make room: close the holes first
            compact_wave(st, c, lane);
            n_hi = st.n_hi[c];
        }
        if ((uint64_t)n_hi + total > (uint64_t)st.cap) status |= ST_CAP_OVERFLOW;
        const int64_t nid = st.next_id[c];
        uint64_t base = 0;
        rank0 = 0;
        for (int i0 = 0; i0 < n_hi && base < (uint64_t)(st.cap - n_hi); i0 += 64) { // (nothing fits beyond: done)
            const int i = i0 + lane;
            const int wc = i < n_hi ? live_class_c(st, cb, i) : -1;
            const uint64_t mlive = __ballot(wc >= 0);
            const uint32_t nd = dups_of(wc, rank0 + (uint32_t)__popcll(mlive & lt)); // (<= cap each: 64 of them fit 32 bits)
            const uint32_t inc = wave_incl_scan_u32(nd, lane);
            const uint64_t first = base + (uint64_t)(inc - nd);
            for (uint32_t g = 0; g < nd; g++) {
                const int64_t dst = (int64_t)n_hi + (int64_t)(first + g);
                if (dst >= (int64_t)st.cap) break; // (flagged above; the id is spent all the same)
                move_particle(st, cb, c, (int)dst, i);
                st.cold[cb + dst].id = nid + (int64_t)(first + g);
            }
            base += (uint64_t)(uint32_t)__shfl(inc, 63, 64);
            rank0 += (uint32_t)__popcll(mlive);
        }
        changed = (uint64_t)n_hi + total > (uint64_t)st.cap ? st.cap - n_hi : (int)total;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0) {
            st.next_id[c] = nid + (int64_t)total;
            st.n_hi[c] = n_hi + changed;
            st.n_part[c] = n_live + changed;
        }
    } else {
        return 0;
    }
    if (lane == 0) {
        mag[cl] = mag[cl] * ratio;
        st.sort_valid[c] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    wave_sync();
    return changed;
}

// aero_state_prepare_weight_for_add [upstream, uncited]: about n_add particles are going to join
// class cl; if that leaves the class outside [n_ideal / 2, 2 n_ideal] its weight is scaled by
// ratio = (n + n_add) / n_ideal so that it comes to n_ideal (pmcx_set_weight_preparation).  Is a
// scaling due -- and would aero_state_scale_weight act on it with the flags as they are?
// cls_cnt: the cell's particles per class, class q in lane q (class_counts: ONE pass over the cell
// for all the modes of a call; kept current by whoever adds or removes -- a count per mode was a
// pass over every particle per mode); n_class == 1: the cell's n_part.
__device__ int class_counts(const DeviceState &st, int c, int lane)
{
    if (!st.prep_on || st.n_class == 1) return 0;
    const int64_t cb = (int64_t)c * st.cap;
    const int n_hi = st.n_hi[c], C = st.n_class;
    int cnt = 0;
    for (int i0 = 0; i0 < n_hi; i0 += 64) {
        const int wc = i0 + lane < n_hi ? live_class_c(st, cb, i0 + lane) : -1;
        for (int q = 0; q < C; q++) {
            const int k = __popcll(__ballot(wc == q));
            if (lane == q) cnt += k;
        }
    }
    return cnt;
}
__device__ __forceinline__ bool weight_preparation_due(const DeviceState &st, int c, int cl, double n_add, int lane,
                                                       double &ratio, int cls_cnt)
{
    if (!st.prep_on) return false;
    const int n_cls = st.n_class == 1 ? st.n_part[c] : __shfl(cls_cnt, cl, 64);
    const double n_new = (double)n_cls + n_add, ideal = st.n_ideal[(int64_t)c * st.n_class + cl];
    if (n_new == 0.0 || !(ideal > 0.0)) return false;
    if (!(n_new < ideal / 2.0 || n_new > ideal * 2.0)) return false;
    ratio = n_new / ideal;
    return (ratio > 1.0 && (st.prep_halving || n_cls == 0)) || (ratio < 1.0 && (st.prep_doubling || n_cls == 0));
}

// appends Poisson(num_conc*factor/magnitude) particles per mode; returns how many
// means != nullptr: the Poisson mean of every mode is given (negative: the mode is skipped)
// Modes m_first .. are taken.  The weight preparation of a mode (weight_preparation_due) is a whole-
// cell pass on the few cells that need one, and its code in every cell's straight path cost the
// scenario update four fifths of its speed: HEAVY = false stops at such a mode (stop_at = its
// index) and the caller hands the cell to the HEAVY = true instance of itself, which scales the
// weight and goes on from there.
template <bool HEAVY>
__device__ int add_dist_sample(const DeviceState &st, int c, const AeroDistDev &d, double factor,
                               uint32_t kind, double create_time, uint64_t seed, uint32_t step,
                               int &status, int lane, const double *means, int m_first, int &stop_at,
                               int &cls_cnt, double char_factor = 1.0)
{
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const double *mag = st.mag + (int64_t)c * st.n_class;
    const int64_t mo = d.mo(c);
    int added = 0;
    stop_at = -1;
    for (int m = m_first; m < d.n_mode; m++) {
        uint32_t w[4];
        const int cl = d.mode_class[mo + m];
        // a negative num_conc: the cell's distribution does not have this mode (a cell without
        // emissions, land under the sea-salt source) -- nothing is drawn, no weight is prepared
        if (!means && d.num_conc[(int64_t)c * d.n_mode + m] < 0.0) continue; // (uniform over the wavefront)
        if (!means) { // adjust the weight if necessary, then the mean at the weight that holds now
            double ratio = 1.0;
            // (for char_factor times the mean: the particles expected over the sampling timescale)
            if (weight_preparation_due(st, c, cl,
                                       (d.num_conc[(int64_t)c * d.n_mode + m] * factor / ld_f64(&mag[cl])) * char_factor, lane,
                                       ratio, cls_cnt)) { // (uniform over the wavefront)
                if (!HEAVY) { stop_at = m; return added; }
                const int ch = scale_weight_wave<true>(st, c, cl, ratio, st.prep_doubling, st.prep_halving, seed, step,
                                                       SCALE_ADD | (kind << 16) | ((uint32_t)m << 8) | (uint32_t)cl, lane,
                                                       status);
                if (lane == cl) cls_cnt += ratio > 1.0 ? -ch : ch; // (thinned, or copies added)
            }
        }
        const double mean = means ? means[m] : d.num_conc[(int64_t)c * d.n_mode + m] * factor / ld_f64(&mag[cl]);
        if (means && mean < 0.0) continue; // (uniform over the wavefront)
        rng_block(seed, cell_id, step, (kind << 8) | (uint32_t)m, SITE_EMIT_N, 0, w);
        int n = poisson(mean, u53(w[0], w[1]));
        int n_hi = st.n_hi[c];
        if (n_hi + n > st.cap && n_hi > st.n_part[c]) { compact_wave(st, c, lane); n_hi = st.n_hi[c]; }
        if (n_hi + n > st.cap) { status |= ST_CAP_OVERFLOW; n = st.cap - n_hi; }
        const int64_t nid = st.next_id[c];
        for (int k = lane; k < n; k += 64) {
            const int i = n_hi + k;
            rng_block(seed, cell_id, step, (kind << 28) | ((uint32_t)m << 20) | (uint32_t)k, SITE_EMIT_P, 0, w);
            const double z = det_norminv(u53(w[2], w[3]));
            const double r = det_exp(d.ln_gm_radius[mo + m] + d.ln_gsd[mo + m] * z);
            const double v = (4.0 / 3.0) * PI * r * r * r;
            double vt = 0.0, ms = 0.0;
            for (int s = 0; s < st.n_spec; s++) {
                const double vs = d.vol_frac[(mo + m) * st.n_spec + s] * v;
                st.vol[(cb + i) * st.n_spec + s] = vs;
                vt = vt + vs;
                ms = ms + st.density[s] * vs;
            }
            for (int s = 0; s < st.n_source; s++)
                norig(st, cb, s, i) = (s == d.mode_source[mo + m]) ? 1 : 0;
            st.hot[cb + i].vtot = vt;
            st.hot[cb + i].mass = ms;
            st.wclass[cb + i] = cl;
            st.gkey[cb + i] = key_of(st, vt, cl);
            st.cold[cb + i].id = nid + k;
            st.hot[cb + i].lct = create_time;
            st.hot[cb + i].gct = create_time;
            comp_single(st, cb + i, d.mode_source[mo + m], create_time);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0) {
            st.next_id[c] = nid + n;
            st.n_hi[c] = n_hi + n;
            st.n_part[c] = st.n_part[c] + n;
        }
        if (lane == cl) cls_cnt += n;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence_block();
        wave_sync();
        added += n;
    }
    return added;
}

// HEAVY = false: one wavefront per cell of the block; a cell whose next mode calls for a weight
// preparation stops there and leaves a record {cell, position, n_emit, n_dil_in, n_dil_out} in
// st.prep_resume.  HEAVY = true: one wavefront per record, from that position on (position: the
// emission mode to go on with, or 512 + the background mode when the dilution is behind it).
template <bool HEAVY>
__device__ __forceinline__ void scenario_cell(const DeviceState &st, int c, int pos, int ne, int ni, int no, double dt,
                                              double create_time, uint64_t seed, uint32_t step,
                                              const AeroDistDev &emission, const double *dilution_rate,
                                              const AeroDistDev &background, int32_t *n_emit, int32_t *n_dil_in,
                                              int32_t *n_dil_out, uint32_t emit_kind, int lane, const double *height,
                                              const double *old_height)
{
    const uint64_t lt = (1ull << lane) - 1ull;
    int status = 0;
  do {
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    // the cell goes to the other instance from position p on
    auto hand_over = [&](int p) {
        if (lane == 0) {
            int32_t *rec = st.prep_resume + 1 + 5 * (int64_t)atomicAdd(&st.prep_resume[0], 1);
            rec[0] = c; rec[1] = p; rec[2] = ne; rec[3] = ni; rec[4] = no;
            if (status) atomicOr(&st.status[c], status);
        }
    };
    apply_event_log(st, c, lane);
    int stop = -1;
    // the class counts (several classes, weight preparation on): a pass over the cell in front of the
    // emission only where the cell HAS an emission mode; everywhere else the dilution's pass, which
    // looks at every particle anyway, leaves them behind for the background's modes
    const bool want_cnt = st.prep_on && st.n_class > 1;
    bool have_cnt = false;
    int cls_cnt = 0;
    if (want_cnt) {
        bool emits = HEAVY;
        for (int m = 0; m < emission.n_mode && pos < 512; m++) emits |= emission.num_conc[(int64_t)c * emission.n_mode + m] >= 0.0;
        if (emits) { cls_cnt = class_counts(st, c, lane); have_cnt = true; }
    }
    // the add's characteristic_factor (wrf_pmc_driver.F90:1794, 1916-1918): sample timescale / step
    const double cf = fmax(1.0, fmin(3600.0, create_time)) / dt;
    if (pos < 512 && emission.n_mode > 0) {
        ne += add_dist_sample<HEAVY>(st, c, emission, dt, emit_kind, create_time, seed, step, status, lane, nullptr, pos, stop,
                                     cls_cnt, cf);
        if (!HEAVY && stop >= 0) { hand_over(stop); break; }
    }
    const double rate = dilution_rate ? dilution_rate[c] : 0.0;
    // p = exp(-rate dt), times old_height / height where the layer grew (as in the gas update)
    const bool grew = height && height[c] > old_height[c] && old_height[c] > 0.0;
    if (rate > 0.0 || grew) {
        double p_keep = det_exp(-rate * dt);
        if (grew) p_keep = p_keep * old_height[c] / height[c];
        const double p_loss = 1.0 - p_keep;
        if (pos < 512) {
            const int n_hi = st.n_hi[c];
            uint32_t rank0 = 0;
            for (int i0 = 0; i0 < n_hi; i0 += 64) {
                const int i = i0 + lane;
                const int wc = i < n_hi ? live_class_c(st, cb, i) : -1;
                const uint64_t mlive = __ballot(wc >= 0);
                bool drop = false;
                if (wc >= 0) {
                    uint32_t w[4];
                    rng_block(seed, cell_id, step, rank0 + (uint32_t)__popcll(mlive & lt), SITE_DILUTE, 0, w);
                    drop = u53(w[0], w[1]) < p_loss;
                }
                if (drop) mark_dead(st, cb, i);
                const uint64_t mdrop = __ballot(drop);
                if (want_cnt && (mdrop || !have_cnt)) // (the class counts follow, or are taken here)
                    for (int q = 0; q < st.n_class; q++) {
                        const int k = have_cnt ? -(int)__popcll(__ballot(drop && wc == q))
                                               : (int)__popcll(__ballot(!drop && wc == q));
                        if (lane == q) cls_cnt += k;
                    }
                no += __popcll(mdrop);
                rank0 += (uint32_t)__popcll(mlive);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_sync();
            if (lane == 0) st.n_part[c] = st.n_part[c] - no;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __threadfence_block();
            wave_sync();
        }
        if (background.n_mode > 0) {
            ni += add_dist_sample<HEAVY>(st, c, background, p_loss, 1u, create_time, seed, step, status, lane, nullptr,
                                         pos >= 512 ? pos - 512 : 0, stop, cls_cnt, cf);
            if (!HEAVY && stop >= 0) { hand_over(512 + stop); break; }
        }
        const int hi = st.n_hi[c], live = st.n_part[c];
        if ((hi - live) * st.compact_den > hi) compact_wave(st, c, lane);
    }
    if (lane == 0) {
        if (n_emit) n_emit[c] = ne;
        if (n_dil_in) n_dil_in[c] = ni;
        if (n_dil_out) n_dil_out[c] = no;
        if (status) atomicOr(&st.status[c], status);
    }
  } while (false);
}

template <bool HEAVY>
__global__ __launch_bounds__(64) void k_scenario_update(DeviceState st, double dt, double create_time,
                                                        uint64_t seed, uint32_t step,
                                                        AeroDistDev emission, const double *dilution_rate,
                                                        AeroDistDev background, int32_t *n_emit,
                                                        int32_t *n_dil_in, int32_t *n_dil_out,
                                                        uint32_t emit_kind, const double *height,
                                                        const double *old_height)
{
    const int lane = threadIdx.x;
    if (!HEAVY) {
        const int c = blockIdx.x;
        if (!in_loop(st, c)) {
            if (lane == 0) {
                if (n_emit) n_emit[c] = 0;
                if (n_dil_in) n_dil_in[c] = 0;
                if (n_dil_out) n_dil_out[c] = 0;
            }
            return;
        }
        scenario_cell<false>(st, c, 0, 0, 0, 0, dt, create_time, seed, step, emission, dilution_rate, background, n_emit,
                             n_dil_in, n_dil_out, emit_kind, lane, height, old_height);
        return;
    }
    const int n_work = st.prep_resume[0];
    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const int32_t *rec = st.prep_resume + 1 + 5 * (int64_t)wi;
        scenario_cell<true>(st, rec[0], rec[1], rec[2], rec[3], rec[4], dt, create_time, seed, step, emission,
                            dilution_rate, background, n_emit, n_dil_in, n_dil_out, emit_kind, lane, height, old_height);
    }
}

// ------------------------------------------------------------ gas scenario
//
// scenario_update_gas_state (call: wrf_pmc_driver.F90:202-205) on every cell: one thread per
// (cell, gas species).  The scenario's series are interpolated at the step's time (constant
// outside, linear inside), emissions become ppb through the cell's air molar density,
// dilution pulls towards the background.  Streams mix_rat once: 16 B per value plus the
// scenario rows it reads.  [UPSTREAM-UNCITED formulas: include/pmcx.h]

__device__ __forceinline__ void interp_find(const double *x, int n, double xq, int &p, double &alpha)
{
    // constant before the first and after the last point, linear in between
    p = 0; alpha = 0.0;
    if (n <= 1 || xq <= x[0]) return;
    if (xq >= x[n - 1]) { p = n - 1; return; }
    while (p < n - 2 && xq >= x[p + 1]) p++;
    alpha = (xq - x[p]) / (x[p + 1] - x[p]);
}
__device__ __forceinline__ double interp_at(const double *y, int64_t stride, int n, int p, double alpha)
{
    if (n <= 1 || (p == 0 && alpha == 0.0)) return y[0];   // (the oracle returns y[0] / y[n-1] there)
    if (p >= n - 1) return y[(int64_t)(n - 1) * stride];
    return (1.0 - alpha) * y[(int64_t)p * stride] + alpha * y[(int64_t)(p + 1) * stride];
}

__global__ __launch_bounds__(256) void k_gas_update(GasScenarioDev g, double *mix_rat,
                                                    const double *temp, const double *pressure,
                                                    const double *height, const double *old_height,
                                                    int n_cells, double dt, double t,
                                                    const uint8_t *loop_mask)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (int64_t)n_cells * g.n_gas) return;
    const int c = (int)(j / g.n_gas), q = (int)(j - (int64_t)c * g.n_gas);
    if (loop_mask && !loop_mask[c]) return;
    double x = mix_rat[j];
    const double h = height[c], h_old = old_height ? old_height[c] : h;
    if (g.n_time_e > 0) {
        int p; double a;
        interp_find(g.emission_time, g.n_time_e, t, p, a);
        const double scale = interp_at(g.emission_rate_scale, 1, g.n_time_e, p, a);
        const double em = interp_at(g.emission + ((int64_t)c * g.n_time_e) * g.n_gas + q, g.n_gas,
                                    g.n_time_e, p, a);
        const double air_molar_den = pressure[c] / (UNIV_GAS * temp[c]);
        const double p_emit = scale * dt / h;
        x = x + p_emit * (em / air_molar_den * 1e9);
    }
    if (g.n_time_d > 0) {
        int p; double a;
        interp_find(g.dilution_time, g.n_time_d, t, p, a);
        const double rate = interp_at(g.dilution_rate + (int64_t)c * g.n_time_d, 1, g.n_time_d, p, a);
        const double bg = interp_at(g.background + ((int64_t)c * g.n_time_d) * g.n_gas + q, g.n_gas,
                                    g.n_time_d, p, a);
        double keep = det_exp(-rate * dt);
        if (h > h_old) keep = keep * h_old / h;
        x = x * keep;
        x = x + bg * (1.0 - keep);
    }
    mix_rat[j] = x < 0.0 ? 0.0 : x;
}

// ----------------------------------------------------------- dry deposition
//
// dry_dep_aero_state (interface/wrf_pmc_dep_aero.F90:154-192) on the cells flagged as
// surface cells: per particle the Zhang (2001) deposition velocity, removal probability
// (dt / dz) vd, one uniform keyed by the particle's rank among the live ones.  One
// workgroup of four wavefronts per cell; removed particles become tombstones.

struct DepArgs {
    const int32_t *surface; // [n_cells] non-zero: the cell touches the ground (NULL: every cell)
    const double *aer_res_a, *ustar, *alpha, *big_a, *height, *inv_density; // [n_cells]
    int32_t *n_removed, *unrealistic;                                       // [n_cells]
};

__global__ __launch_bounds__(256) void k_dry_deposition(DeviceState st, DepArgs a, double dt,
                                                        uint64_t seed, uint32_t step)
{
    __shared__ uint32_t s_live[4];
    __shared__ int s_nd, s_bad;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = blockIdx.x;
    if (a.surface && !a.surface[c]) { // (the whole workgroup)
        if (t == 0) { a.n_removed[c] = 0; a.unrealistic[c] = 0; }
        return;
    }
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << lane) - 1ull;
    const double ra = a.aer_res_a[c], ustar = a.ustar[c], alpha = a.alpha[c], big_a = a.big_a[c];
    const double rrho = a.inv_density[c], temp = st.temp[c], dt_dz = dt / a.height[c];
    const int n_hi = st.n_hi[c];
    if (wave == 0) apply_event_log(st, c, lane);
    __syncthreads();
    // Each wavefront takes a contiguous quarter of the slots (a multiple of 64).  A particle's
    // uniform is keyed by its rank among the live ones: the quarters are counted first.
    const int per = (((n_hi + 3) >> 2) + 63) & ~63;
    const int lo = min(n_hi, wave * per), hi = min(n_hi, lo + per);
    uint32_t cnt = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        cnt += (uint32_t)__popcll(__ballot(i < hi && live_class_c(st, cb, i) >= 0));
    }
    if (lane == 0) s_live[wave] = cnt;
    if (t == 0) { s_nd = 0; s_bad = 0; }
    __syncthreads();
    uint32_t rank0 = 0;
    for (int w = 0; w < wave; w++) rank0 += s_live[w];
    int nd = 0, bad = 0;
    // Almost every particle stays: its uniform lies above (dt/dz) (vs + 1/ra), a bound of its
    // removal probability that costs a dozen single-precision operations instead of the six
    // hundred of the deposition velocity (vd = vs + 1/(ra + rs + ra rs vs) <= vs + 1/ra for
    // positive resistances; vs grows with the diameter and its slip factor is at most
    // 1 + 3.314 lambda / d; every factor rounded up with room to spare).  The particles the bound
    // does not settle wait in a queue and are evaluated 64 at a time, all lanes busy -- the
    // same value for each as if it had been evaluated in place.  With unusual inputs (a
    // non-positive resistance, ...) nothing is screened: the flag for unrealistic probabilities
    // then sees every particle, as the reference's loop does.
    __shared__ int32_t s_qslot[4][128];
    __shared__ double s_qu[4][128];
    int queued = 0; // (uniform within the wavefront)
    const bool sane = ra > 0.0 && ustar > 0.0 && alpha > 0.0 && big_a > 0.0 && rrho > 0.0 &&
                      temp > 0.0 && dt_dz > 0.0;
    const float inv_ra_up = sane ? (float)(1.0 / ra) * 1.0001f : 0.0f;
    const float dtdz_up = sane ? (float)dt_dz * 1.0001f : 0.0f;
    auto evaluate = [&](int take) { // the first `take` of the queue
        const bool on = lane < take;
        const int i = on ? s_qslot[wave][lane] : 0;
        const double u = on ? s_qu[wave][lane] : 0.0;
        bool drop = false;
        if (on) {
            const double v = st.hot[cb + i].vtot, m = st.hot[cb + i].mass;
            const double diam = 2.0 * det_cbrt(v * (0.75 / PI));
            const double vd = dep_velocity(diam, m / v, temp, rrho, ustar, ra, alpha, big_a);
            const double prob = dt_dz * vd;
            bad |= (prob >= 1.0) | (prob <= 0.0);
            drop = u < prob;
        }
        if (drop) mark_dead(st, cb, i);
        nd += __popcll(__ballot(drop));
        lds_sync();
        const int rest = queued - take;
        int rs = 0;
        double ru = 0.0;
        if (lane < rest) { rs = s_qslot[wave][take + lane]; ru = s_qu[wave][take + lane]; }
        lds_sync();
        if (lane < rest) { s_qslot[wave][lane] = rs; s_qu[wave][lane] = ru; }
        queued = rest;
        lds_sync();
    };
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < hi && live_class_c(st, cb, i) >= 0;
        const uint64_t mlive = __ballot(live);
        bool need = false;
        double u = 0.0;
        if (live) {
            uint32_t w[4];
            rng_block(seed, cell_id, step, rank0 + (uint32_t)__popcll(mlive & lt), SITE_DEPOSIT, 0, w);
            u = u53(w[0], w[1]);
            need = true;
            const double v = st.hot[cb + i].vtot, m = st.hot[cb + i].mass;
            if (sane && v > 0.0 && m > 0.0) {
                const float vf = (float)v, mf = (float)m;
                // d_up >= 2 cbrt(v 0.75 / pi), dens_up >= m / v
                const float d_up = 2.0002f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(vf * 0.238732415f) * 0.333333343f);
                const float dens_up = (mf / vf) * 1.0001f;
                // vs <= dens g (d^2 + 3.314 lambda d) / (18 mu)
                const float vs_up = dens_up * (float)(GRAV / (18.0 * AIR_DYN_VISC)) * 1.0001f *
                                    (d_up * d_up + 2.1542e-7f * d_up);
                const float prob_up = dtdz_up * (vs_up + inv_ra_up) * 1.001f;
                need = !(u >= (double)prob_up);
            }
        }
        const uint64_t mneed = __ballot(need);
        if (need) {
            const int pos = queued + __popcll(mneed & lt);
            s_qslot[wave][pos] = i; s_qu[wave][pos] = u;
        }
        queued += __popcll(mneed);
        lds_sync();
        if (queued >= 64) evaluate(64);
        rank0 += (uint32_t)__popcll(mlive);
    }
    if (queued > 0) evaluate(queued);
    bad = __ballot(bad != 0) != 0;
    if (lane == 0) {
        if (nd) atomicAdd(&s_nd, nd);
        if (bad) atomicOr(&s_bad, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
    if (t == 0) {
        st.n_part[c] = st.n_part[c] - s_nd;
        if (s_nd) st.sort_valid[c] = 0;
        a.n_removed[c] = s_nd;
        a.unrealistic[c] = s_bad;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
    if (wave == 0) { // (compaction is one wavefront's work)
        const int top = st.n_hi[c], live_n = st.n_part[c];
        if ((top - live_n) * st.compact_den > top) compact_wave(st, c, lane);
    }
}

// ---------------------------------------------------------------- transport
//
// Stochastic particle transport between the cells of a block:
// aero_state_multisample + aero_state_sample_particles_by_class
// (interface/wrf_pmc_trans_aero.F90:2351-2482, 2181-2322) in per-particle form
// (every particle draws its own destination; see oracle/pmco.c for the law).
//   k_trans_decide  one wavefront per source cell: destination tables of each
//                   class in LDS (sorted by probability, cumulative thresholds,
//                   weight ratios), an 8-bit prefix of the destination uniform per live particle
//                   (sixteen particles to a Philox block, addressed by rank) that settles who
//                   stays; the few it cannot tell for are queued and finished 64 at a time (the
//                   rest of the uniform and the copies from a block of their own); movers
//                   compacted into the cell's outbox in rank order; a mover's sort key is
//                   marked dead here
//   k_trans_gather  one wavefront per destination cell: scans the outboxes of
//                   its sources in a fixed order, appends the arrivals and leaves each
//                   mover's tombstone in the record it has just read
//   k_trans_remove  one wavefront per source cell: the cell's live count, and the
//                   tombstones of the movers that no gather of this block reads

constexpr int TRANS_STAYS = (int)0x80000000u; // outbox entry: the source particle stays (an emitter's copies)
// outbox entry of a mover that also emits (law 1, where the two intervals overlap): its record is
// read by the gathers of its copies' destinations too, so the gather of its own destination leaves
// it alone and k_trans_remove writes the tombstone
constexpr int TRANS_KEEP = 0x40000000;
constexpr int TRANS_COPIES_MASK = 0x3FFF;
constexpr int TRANS_MAX_COPIES = 4096; // (a guard, not a law: P(G > 4096) is 2e-18 at r = 0.01, 1.7 % of the mean at r = 0.001; law 1: flagged ST_TRANS_COPIES)
constexpr uint32_t TRANS_BOTH_BLOCK = 1u << 20; // law 1: block of the copy count of a particle that moves AND emits
constexpr double TRANS_EMIT_WALK_MAX = 32.0;    // law 1: above this sum of rates the zero-truncation is below 2^-46: plain Poisson

// law 1: the number of copies an emitter sends, a zero-truncated Poisson(sum_mu) from one uniform
// (the walk's first term exp(-sum_mu) sum_mu / (1 - exp(-sum_mu)) underflows for large rates, where the
// truncation no longer matters: the plain count, at least one)
// (not inlined, like deal_copies below: the emitters are a few particles in a hundred, and their code and
// registers in k_trans_decide's straight path cost every particle -- 1.80 -> 2.3 ms at 1.2e9 particles)
__device__ __attribute__((noinline)) int emit_count(double sum_mu, double u2, bool &clamped)
{
    int n;
    if (sum_mu > TRANS_EMIT_WALK_MAX) {
        n = poisson(sum_mu, u2);
        if (n < 1) n = 1;
    } else {
        const double e0 = det_exp(-sum_mu);
        double term = e0 * sum_mu / (1.0 - e0), cdf = term;
        n = 1;
        while (u2 >= cdf && n < TRANS_MAX_COPIES) {
            n++;
            term = term * sum_mu / (double)n;
            const double next = cdf + term;
            if (next == cdf) break;   // (the sum has stopped short of a uniform within rounding of 1)
            cdf = next;
        }
    }
    if (n > TRANS_MAX_COPIES) { clamped = true; n = TRANS_MAX_COPIES; }
    return n;
}

// law 1: an emitter's n_emit copies dealt to the r < 1 destinations of its class in proportion to their
// rates (mucum: the running sums in table order, -1 where r >= 1; the uniform of copy q from the particle's
// blocks 2 + q / 2).  One copy: returns its destination.  Several: deal[y] counts them per destination,
// n_dst = how many destinations take any.
__device__ __attribute__((noinline)) int deal_copies(uint64_t seed, uint32_t cell_id, uint32_t step, uint32_t rank,
                                                     int n_emit, double sum_mu, const double *mucum, int nt,
                                                     uint16_t *deal, uint32_t &n_dst)
{
    int d_one = -1;
    n_dst = 0;
    if (n_emit > 1) for (int y = 0; y < nt; y++) deal[y] = 0;
    for (int q = 0; q < n_emit; q++) {
        uint32_t v[4];
        rng_block(seed, cell_id, step, rank, SITE_TRANS, 2u + (uint32_t)(q >> 1), v);
        const double xq = u53(v[2 * (q & 1)], v[2 * (q & 1) + 1]) * sum_mu;
        int d = -1, last = -1;
        for (int y = 0; y < nt; y++) {
            const double cm = mucum[y];   // (the sum the oracle forms term by term, same order)
            if (cm < 0.0) continue;
            last = y;
            if (xq < cm) { d = y; break; }
        }
        if (d < 0) d = last;
        if (n_emit == 1) { d_one = d; n_dst = 1; }
        else { if (deal[d] == 0) n_dst++; deal[d] = (uint16_t)(deal[d] + 1); }
    }
    return d_one;
}

// open lateral boundaries: is local row j the domain's first or last row?
__device__ __forceinline__ bool lateral_row(const DeviceState &st, int j, int ny)
{
    const int gj = j + st.trans_j0, gny = st.trans_gny ? st.trans_gny : ny;
    return gj == 0 || gj == gny - 1;
}

// ... and is local column i the domain's first or last column?  (a block cut in x carries ghost
// columns: its column 0 stands for the domain's column trans_i0)
__device__ __forceinline__ bool lateral_col(const DeviceState &st, int i, int nx)
{
    const int gi = i + st.trans_i0, gnx = st.trans_gnx ? st.trans_gnx : nx;
    return gi == 0 || gi == gnx - 1;
}
// a ghost column of a block cut in x
__device__ __forceinline__ bool ghost_col(const DeviceState &st, int i, int nx)
{
    return st.trans_cut_x && (i == 0 || i == nx - 1);
}

struct TransGrid {
    int32_t nx, nz, ny, periodic_x, periodic_y;
    const double *prob_h, *prob_v, *cell_volume;
    int2 *outbox;     // [n_cells][cap]  {slot | copies << 16, destination cell}
    int32_t *n_out;   // [n_cells]
    int32_t *n_arr;   // [n_cells]
    int32_t c_first;  // gather / remove work on cells [c_first, c_first + gridDim.x): the real rows
                      // of a slab whose first and last rows are ghosts of the neighbouring ranks
    int32_t *n_exp;   // [n_cells] arrivals a cell is about to receive (slots), counted by the senders
    uint64_t *dest_mask; // [n_cells] which of its neighbours a cell sends to this step: bit 0 east,
                         // 1 west, 2 north, 3 south, 4 + k level k of its own column (all ones when
                         // the grid is too small or too tall for the code to be unambiguous)
};

// EMIT: law 1 (pmcx_set_transport_law) -- its own instance, so that the emitters' code and registers
// (82 against 45) are not the default law's to carry (1.73 -> 1.99 ms when they were)
#ifndef PMCX_DECIDE_WAVES
#define PMCX_DECIDE_WAVES 6
#endif
// (six wavefronts a SIMD: the kernel streams keys and waits for them -- at 88 registers, five wavefronts,
// the law-1 instance ran 2.3 ms where round 3's 82 registers gave 1.8)
template <bool EMIT>
__global__ __launch_bounds__(64, PMCX_DECIDE_WAVES) void k_trans_decide(DeviceState st, TransGrid tg, uint64_t seed,
                                                     uint32_t step)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x, c = blockIdx.x;
    const int C = st.n_class, nx = tg.nx, nz = tg.nz, ny = tg.ny;
    const int ND = 4 + nz;
    // The tables of a class cost 33 bytes per destination (60 until round 4: with ten weight classes six
    // wavefronts shared a CU's LDS where twenty-four stream keys with one): the thresholds' lower ends
    // are the upper ends in front of them, the destination's bit of the mask is kept as its index, the
    // scratch of the sort is one class's.  (Pointers formed from smem and offsets: through an integer a
    // pointer would lose its address space and every access of the mover's path become a FLAT one, whose
    // wait is for all the wavefront's requests, the next trip's keys among them.)
    double *t_hi = (double *)smem;           // [C][ND] cumulative thresholds (their upper ends)
    double *t_ratio = t_hi + C * ND;         // [C][ND]
    // law 1: the running sum of the rates over a class's r < 1 destinations in table order (-1: r >= 1):
    // what a copy's uniform is compared with when it is dealt (formed once per cell, not per copy)
    double *t_mucum = t_ratio + C * ND;      // [C][ND]
    double *t_p = t_mucum + C * ND;          // [ND] scratch: a class's probabilities before sorting
    // law 1, per class: the sum of the emitters' rates, their interval [s_act_lo, s_act_hi) of the placing uniform
    double *s_mu_sum = t_p + ND;
    double *s_act_lo = s_mu_sum + C, *s_act_hi = s_act_lo + C;
    int *t_dest = (int *)(s_act_hi + C);     // [C][ND]
    int *t_cnt = t_dest + C * ND;            // [C][ND] copies sent to each destination
    int *t_tmpd = t_cnt + C * ND;            // [ND] scratch: a class's destinations before sorting
    int *t_n = t_tmpd + ND;                  // [C]
    // law 1: a lane's copies per destination while it deals them ([64][ND], an emitter of several copies)
    uint16_t *s_deal = (uint16_t *)(t_n + C);
    uint8_t *t_bit = (uint8_t *)(s_deal + 64 * ND); // [C][ND] the destination's bit of dest_mask (255: every bit)
    auto bit_of = [](uint8_t b) -> uint64_t { return b == 255u ? ~0ull : 1ull << b; };
    for (int x = lane; x < C * ND; x += 64) t_cnt[x] = 0;
    const int i = c % nx, k = (c / nx) % nz, j = c / (nx * nz);
    const bool mask_ok = ND <= 64 && (nx >= 3 || !tg.periodic_x) && (ny >= 3 || !tg.periodic_y);
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << lane) - 1ull;
    apply_event_log(st, c, lane); // (this step's removals and arrivals keep the cell's keys current)

    // destination tables, class by class: the candidates (four neighbours, then the levels) are
    // looked at one per lane and kept in that order; every kept one finds its place in the stable
    // ascending order of the probabilities by counting (n compares per lane instead of n^2 on
    // one); its weight ratio and its bit of the destination mask are its own lane's work; only
    // the cumulative thresholds are taken in sequence (their sums depend on the order).
    for (int cl = 0; cl < C; cl++) {
        double *p = t_p;
        int *dc = t_tmpd;
        int *ord = t_cnt + cl * ND; // rank -> element (t_cnt is cleared again below)
        const bool bx = lateral_col(st, i, nx), by = lateral_row(st, j, ny);
        const bool lateral_cell = st.trans_lateral && (bx || by);
        int n = 0;
        for (int d0 = 0; d0 < ND; d0 += 64) {
            const int d = d0 + lane;
            double pr = 0.0;
            int dcell = 0;
            if (d < ND) {
                if (lateral_cell) {
                    // open lateral boundaries (wrf_pmc_trans_aero.F90:689-768): a boundary cell only
                    // sends inwards, with min(prob, 1); a corner sends nothing
                    const int gi0 = i + st.trans_i0, gnx0 = st.trans_gnx ? st.trans_gnx : nx;
                    const int din = gi0 == 0 ? 1 : gi0 == gnx0 - 1 ? 0 : (j + st.trans_j0 == 0) ? 3 : 2;
                    if (!(bx && by) && d == din) {
                        pr = tg.prob_h[((int64_t)c * C + cl) * 4 + d];
                        if (pr > 1.0) pr = 1.0;
                        dcell = ((j + (d == 2 ? -1 : d == 3 ? 1 : 0)) * nz + k) * nx + (i + (d == 0 ? -1 : d == 1 ? 1 : 0));
                    }
                } else if (d < 4) {
                    int ii = i + (d == 0 ? -1 : d == 1 ? 1 : 0), jj = j + (d == 2 ? -1 : d == 3 ? 1 : 0);
                    bool ok = true;
                    if (ii < 0 || ii >= nx) { if (!tg.periodic_x) ok = false; else ii = (ii + nx) % nx; }
                    if (jj < 0 || jj >= ny) { if (!tg.periodic_y) ok = false; else jj = (jj + ny) % ny; }
                    if (ok) { pr = tg.prob_h[((int64_t)c * C + cl) * 4 + d]; dcell = (jj * nz + k) * nx + ii; }
                } else {
                    const int kk = d - 4;
                    if (kk != k) { pr = tg.prob_v[((int64_t)c * C + cl) * nz + kk]; dcell = (j * nz + kk) * nx + i; }
                }
            }
            const bool keep = pr > 0.0;
            const uint64_t mk = __ballot(keep);
            if (keep) { const int at = n + __popcll(mk & lt); p[at] = pr; dc[at] = dcell; }
            n += __popcll(mk);
        }
        lds_sync();
        for (int x = lane; x < n; x += 64) {
            const double px = p[x];
            int rk = 0;
            for (int y = 0; y < n; y++) { const double py = p[y]; rk += (py < px) || (py == px && y < x); }
            ord[rk] = x;
        }
        lds_sync();
        const double mag_from = st.mag[(int64_t)c * C + cl];
        for (int r = lane; r < n; r += 64) {
            const int a = ord[r], dca = dc[a];
            t_dest[cl * ND + r] = dca;
            const double factor = tg.cell_volume[c] / tg.cell_volume[dca];
            t_ratio[cl * ND + r] = (st.mag[(int64_t)dca * C + cl] / factor) / mag_from;
            // where the destination lies, as the gather of that cell will look for it
            const int di_ = dca % nx, dk_ = (dca / nx) % nz, dj_ = dca / (nx * nz);
            uint8_t bit = 255u;
            if (mask_ok) {
                if (di_ == i && dj_ == j) bit = (uint8_t)(4 + dk_);
                else if (dj_ == j) bit = (di_ == (i + 1) % nx) ? 0u : 1u;
                else bit = (dj_ == (j + 1) % ny) ? 2u : 3u;
            }
            t_bit[cl * ND + r] = bit;
            t_hi[cl * ND + r] = p[a];   // (the probability, in table order: summed below, a lane per class)
        }
        if (lane == 0) t_n[cl] = n;
        lds_sync();
    }
    for (int x = lane; x < C * ND; x += 64) t_cnt[x] = 0;
    // law 1 (pmcx_set_transport_law): behind the movers' intervals one more, the emitters' -- a
    // particle that sends copies towards the destinations with r < 1, rate
    // mu_d = p_d (1 - r_d) / r_d each (the non-removing picks of the reference's Poisson sampler,
    // wrf_pmc_trans_aero.F90:2236-2266); width w = 1 - exp(-sum mu).  Where the two do not fit side
    // by side (sum p + w > 1: a large probability to leave and a small weight ratio) the emitters'
    // interval is [1 - w, 1) and overlaps the movers': a particle in the overlap moves AND emits, as
    // a particle of the reference can be picked without removal before the pick that removes it.
    // Either way P(moves to d) = p_d and E[copies to d] = mu_d: arrivals n p_d / r_d in the mean.
    for (int cl = lane; cl < C; cl += 64) {
        const int nt = t_n[cl];
        // the cumulative thresholds, taken in sequence (their sums depend on the order)
        double sum_mu = 0.0;
        {
            double rem = 1.0, cum = 0.0;
            for (int d = 0; d < nt; d++) {
                const double pa = t_hi[cl * ND + d], lo = cum;
                if (rem >= 1e-8) {
                    cum = cum + pa;
                    if (cum > 1.0) cum = 1.0;
                    rem = rem - pa;
                }
                t_hi[cl * ND + d] = cum;
                if (EMIT) {
                    const double r = t_ratio[cl * ND + d];
                    if (r < 1.0) sum_mu = sum_mu + (cum - lo) * (1.0 - r) / r;
                    t_mucum[cl * ND + d] = r < 1.0 ? sum_mu : -1.0;
                }
            }
        }
        double a_lo = nt > 0 ? t_hi[cl * ND + nt - 1] : 0.0, a_hi = a_lo;
        if (EMIT && nt > 0) {
            const double w_emit = 1.0 - det_exp(-sum_mu);
            a_hi = a_lo + w_emit;
            if (a_hi > 1.0) { a_hi = 1.0; a_lo = 1.0 - w_emit; }
        }
        s_mu_sum[cl] = sum_mu; s_act_lo[cl] = a_lo; s_act_hi[cl] = a_hi;
    }
    lds_sync();

    const int n_hi = st.n_hi[c];
    int2 *box = tg.outbox + cb;
    uint32_t rank0 = 0, n_out = 0;
    uint64_t bits = 0;
    // Liveness and class of a slot come from its 2-byte sort key while the cell's keys hold
    // (the log was applied above): a dense plane instead of one field of the 32-byte records.
    const bool keyed = st.key_valid[c] != 0;
    const uint16_t *gk = st.gkey + cb;
    const int Bn = st.n_bin;
    const float inv_bn = 1.0f / (float)Bn;
    auto slot_class = [&](int sl) -> int {
        if (sl >= n_hi) return -1;
        if (!keyed) return live_class(st, cb, sl);
        const uint32_t kq = gk[sl];
        if (kq == 0xFFFFu) return -1;
        if (C == 1) return 0;
        // kq / Bn without the integer division (keys are below 65 536: the estimate is off by one at most)
        uint32_t q = (uint32_t)((float)kq * inv_bn);
        if (q * (uint32_t)Bn > kq) q--;
        else if ((q + 1u) * (uint32_t)Bn <= kq) q++;
        return (int)q;
    };
    // (one class: the stay-or-go threshold is the same for every particle of the cell)
    const double hi_last0 = s_act_hi[0];
    __shared__ __align__(16) uint32_t s_words[256];
    __shared__ uint32_t s_qent[128], s_qrank[128]; // particles the prefix cannot tell for: slot | prefix << 16 | class << 24, rank
    // The destination uniform of the particle of rank r starts with an 8-bit prefix: byte r & 15 of
    // the Philox block addressed by r >> 4 -- sixteen particles to a block.  prefix / 256 at or above
    // the class's total probability of leaving says "stays" (u >= prefix / 256), which settles all
    // but a few per cent of the particles; one the prefix cannot tell for draws the rest from a
    // block of its own, u = (prefix + w0 2^-32) / 256, and from the same block the uniform of its
    // copies.  64 blocks (one per lane, side by side in LDS) cover 1024 ranks, four trips of 4 x 63
    // slots; the undecided are queued in rank order and evaluated 64 at a time, dense lanes.
    // (the keys of the next trip are requested before this trip's work: a wavefront alone with
    // its cell would otherwise wait out a memory round trip forty times)
    auto key_of_slot = [&](int sl) -> uint32_t { return (lane < 63 && sl < n_hi) ? (uint32_t)gk[sl] : 0xFFFFu; };
    auto class_of_key = [&](uint32_t kq) -> int {
        if (kq == 0xFFFFu) return -1;
        if (C == 1) return 0;
        uint32_t q = (uint32_t)((float)kq * inv_bn);
        if (q * (uint32_t)Bn > kq) q--;
        else if ((q + 1u) * (uint32_t)Bn <= kq) q++;
        return (int)q;
    };
    int queued = 0; // entries in the queue (uniform)
    bool out_full = false; // law 1: the outbox ran out of entries (uniform)
    bool copies_clamped = false; // law 1: an emitter's count met the guard
    // the first `take` queued particles: the rest of their uniform, destination, copies, outbox
    auto evaluate = [&](int take) {
        const uint32_t ent = lane < take ? s_qent[lane] : 0u, rank = lane < take ? s_qrank[lane] : 0u;
        const int s = (int)(ent & 0xFFFFu), cl = (int)(ent >> 24);
        bool moves = false;
        int dest = 0, copies = 0, dslot = 0;
        int n_emit = 0;   // law 1: copies this lane's particle sends (while it stays, or before it goes)
        if (lane < take) {
            uint32_t w[4];
            rng_block(seed, cell_id, step, rank, SITE_TRANS, 1, w);
            const double u = ((double)((ent >> 16) & 0xFFu) + (double)w[0] * 0x1.0p-32) * 0x1.0p-8;
            const int nt = t_n[cl];
            if (EMIT && nt > 0 && !(u < s_act_lo[cl]) && u < s_act_hi[cl]) {
                // an emitter: a zero-truncated Poisson(sum mu) number of copies; one that also moves
                // (the overlap) takes the count's uniform from a block of its own -- the mover's
                // copies have the first one's
                double u2 = u53(w[2], w[3]);
                if (u < t_hi[cl * ND + nt - 1]) {
                    uint32_t v[4];
                    rng_block(seed, cell_id, step, rank, SITE_TRANS, TRANS_BOTH_BLOCK, v);
                    u2 = u53(v[0], v[1]);
                }
                bool clamped = false;
                n_emit = emit_count(s_mu_sum[cl], u2, clamped);
                if (clamped) copies_clamped = true;
            }
            if (nt > 0 && u < t_hi[cl * ND + nt - 1]) {
                // the first threshold above u, looked for from the top: the probabilities are in
                // ascending order, the last few destinations hold most of the mass
                int d = nt - 1;
                while (d > 0 && u < t_hi[cl * ND + d - 1]) d--;
                const double r = t_ratio[cl * ND + d];
                copies = 1;
                if (r != 1.0) {
                    const double u2 = u53(w[2], w[3]);
                    if (r > 1.0) copies = (u2 < 1.0 / r) ? 1 : 0;
                    else if (!EMIT) {
                        copies = 1 + (int)(det_log(1.0 - u2) / det_log(1.0 - r));
                        if (copies > TRANS_MAX_COPIES) copies = TRANS_MAX_COPIES;
                    } // (law 1: a mover arrives once; the other 1 / r - 1 come from the emitters)
                }
                dslot = cl * ND + d;
                dest = t_dest[dslot];
                bits |= bit_of(t_bit[dslot]);
                moves = true;
            }
        }
        if (!EMIT) {
            const uint64_t mm = __ballot(moves);
            if (moves) box[n_out + (uint32_t)__popcll(mm & lt)] = make_int2(s | (copies << 16), dest);
            n_out += (uint32_t)__popcll(mm);
        } else {
            // entries of the chunk in rank order: a mover's one, then an emitter's -- every copy dealt to
            // a destination with r < 1 in proportion mu (its uniform from the particle's own blocks),
            // one entry per destination that takes any, in table order, with its count: what a
            // destination gathers from the cell is the same sequence of records, under the same new
            // ids, as one entry per copy in the order dealt would give it
            const int nt = t_n[cl];
            uint16_t *deal = s_deal + lane * ND;
            uint32_t n_dst = 0;  // destinations that take copies
            int d_one = -1;      // the destination of a single copy
            if (n_emit > 0)
                d_one = deal_copies(seed, cell_id, step, rank, n_emit, s_mu_sum[cl], t_mucum + cl * ND, nt, deal, n_dst);
            const uint32_t ne = (moves ? 1u : 0u) + n_dst;
            const uint32_t inc = wave_incl_scan_u32(ne, lane);
            const uint32_t first = n_out + inc - ne;
            // (an outbox holds cap entries: the first particle whose entries do not fit closes it -- its
            // entries and every later particle's are dropped, a mover among them stays -- and is flagged)
            const bool fits = !out_full && first + ne <= (uint32_t)st.cap;
            if (moves && fits) box[first] = make_int2(s | (copies << 16) | (n_emit > 0 ? TRANS_KEEP : 0), dest);
            if (n_emit > 0 && fits) {
                uint32_t at = first + (moves ? 1u : 0u);
                for (int y = (n_emit == 1 ? d_one : 0); y < (n_emit == 1 ? d_one + 1 : nt); y++) {
                    const int k_y = n_emit == 1 ? 1 : (int)deal[y];
                    if (k_y == 0) continue;
                    box[at++] = make_int2(s | (k_y << 16) | TRANS_STAYS, t_dest[cl * ND + y]);
                    bits |= bit_of(t_bit[cl * ND + y]);
                    atomicAdd(&t_cnt[cl * ND + y], k_y);
                }
            }
            if (!fits) moves = false;
            const uint64_t nofit = __ballot(ne > 0 && !fits);
            if (nofit) { // (uniform)
                if (!out_full) n_out = (uint32_t)__shfl((int)first, __builtin_ctzll(nofit), 64);
                out_full = true;
            } else {
                n_out += (uint32_t)__shfl(inc, 63, 64);
            }
        }
        // (a mover's sort key says "gone" from here on; its record lives until the gather of its
        // destination, or k_trans_remove, has dealt with it)
        if (moves && keyed) st.gkey[cb + s] = (uint16_t)0xFFFFu;
        if (moves && copies > 0) atomicAdd(&t_cnt[dslot], copies);
        // what is left moves to the front of the queue
        lds_sync();
        const int rest = queued - take;
        uint32_t re = 0, rr = 0;
        if (lane < rest) { re = s_qent[take + lane]; rr = s_qrank[take + lane]; }
        lds_sync();
        if (lane < rest) { s_qent[lane] = re; s_qrank[lane] = rr; }
        queued = rest;
        lds_sync();
    };
    uint32_t knext[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
    if (keyed) {
#pragma unroll
        for (int ch = 0; ch < 4; ch++) knext[ch] = key_of_slot(ch * 63 + lane);
    }
    uint32_t B0 = 0, covered = 0; // the blocks in LDS cover ranks [16 B0, covered)
    for (int i0 = 0; i0 < n_hi; i0 += 4 * 63) {
        int cls[4];
        if (keyed) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) cls[ch] = class_of_key(knext[ch]);
#pragma unroll
            for (int ch = 0; ch < 4; ch++) knext[ch] = key_of_slot(i0 + 4 * 63 + ch * 63 + lane);
        } else {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) cls[ch] = lane < 63 ? slot_class(i0 + ch * 63 + lane) : -1;
        }
        if (rank0 + 4u * 63u > covered) { // (uniform) the next 64 blocks, one per lane
            B0 = rank0 >> 4;
            covered = 16u * (B0 + 64u);
            uint32_t wb[4];
            rng_block(seed, cell_id, step, B0 + (uint32_t)lane, SITE_TRANS, 0, wb);
            lds_sync(); // (the previous trips' reads are done)
            *(uint4 *)&s_words[4 * lane] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
            lds_sync();
        }
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            const int s = i0 + ch * 63 + lane, cl = cls[ch];
            const uint64_t mlive = __ballot(cl >= 0);
            const uint32_t rank = rank0 + (uint32_t)__popcll(mlive & lt);
            const uint32_t idx = (rank - 16u * B0) & 1023u;
            const uint32_t p8 = (s_words[idx >> 2] >> (8u * (idx & 3u))) & 0xFFu;
            // (one class: no table look-up for the particles that stay, almost all of them)
            const bool need = cl >= 0 && (double)p8 * 0x1.0p-8 < (C == 1 ? hi_last0 : s_act_hi[cl]);
            const uint64_t mneed = __ballot(need);
            if (need) {
                const int pos = queued + __popcll(mneed & lt);
                s_qent[pos] = (uint32_t)s | (p8 << 16) | ((uint32_t)cl << 24);
                s_qrank[pos] = rank;
            }
            queued += __popcll(mneed);
            rank0 += (uint32_t)__popcll(mlive);
            if (mneed) lds_sync();
            if (queued >= 64) evaluate(64);
        }
    }
    if (queued > 0) evaluate(queued);
    // the arrivals each destination has to expect: one atomic per destination, not per mover
    wave_sync();
    for (int x = lane; x < C * ND; x += 64)
        if (x % ND < t_n[x / ND] && t_cnt[x] > 0) {
            const int dcell = t_dest[x], di_ = dcell % nx, dj_ = dcell / (nx * nz);
            // (what is sent to an open lateral boundary cell leaves the domain: nobody gathers it)
            if (st.trans_lateral && (lateral_col(st, di_, nx) || lateral_row(st, dj_, ny))) continue;
            atomicAdd(&tg.n_exp[dcell], t_cnt[x]);
        }
    for (int d = 32; d >= 1; d >>= 1) bits |= __shfl_xor(bits, d, 64);
    if (__ballot(copies_clamped) != 0 && lane == 0) atomicOr(&st.status[c], ST_TRANS_COPIES);
    if (lane == 0) {
        tg.n_out[c] = (int32_t)n_out; tg.dest_mask[c] = bits;
        if (out_full) atomicOr(&st.status[c], ST_OUTBOX_OVERFLOW);
    }
}

// The transport probabilities of one step, built where the particles are (SURVEY 8 f2,
// wrf_pmc_trans_driver's first half, interface/wrf_pmc_trans.F90:95-156): one workgroup per
// column (i, j).
//   phase 0  particles per class of every level (aero_state_total_particles): the live count
//            with one class, else the 2-byte sort keys while they hold, else the records
//   phase 1  compute_advect_probs_wrf (:205-286) and compute_diffusion_probs (:358-409), one
//            thread per (level, class) / (level, direction)
//   phase 2  stable_timestep (:629-651), the tridiagonal B of compute_vertical_probs (:414-492)
//            and B* = B^N with the reference's row normalisation after every product (:494-504):
//            B* dense in LDS, one thread per element, three terms (B is tridiagonal) added in
//            ascending inner index, row sums by one thread per row in element order
//   phase 3  per class R = B* A (:507-530), then normalize_probs (:692-773) per level
// Arithmetic is the reference's, operation for operation (single-precision where grid% is).
struct TransFieldsDev {
    int32_t nx, nz, ny, do_vertmix, all_columns, i_offset, j_offset, global_nx, global_ny;
    double dt, dx, dy;
    const float *alt, *xkhh, *u_flux, *v_flux, *w_flux, *z, *z_at_w, *exch_h;
    double *prob_h, *prob_v;
    int32_t *n_substeps, *bad, *n_class_part;
    int32_t *clamp_flag; // [1] set when some column asked for more than TRANS_MAX_SUBSTEPS sub-steps
};

// stable_timestep's count (wrf_pmc_trans.F90:629-651) is unbounded in the reference; a degenerate
// column (z(2) == z(1), a huge exch_h) would turn one launch into seconds of serial matrix steps.
// The count is clamped here, the column flagged (bit 16 of bad) and the call fails.
constexpr double TRANS_MAX_SUBSTEPS = 1.0e4;

__global__ __launch_bounds__(256) void k_trans_probs(DeviceState st, TransFieldsDev f)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nx = f.nx, nz = f.nz, ny = f.ny, C = st.n_class;
    const int i = blockIdx.x % nx, j = blockIdx.x / nx;
    double *Bt = (double *)smem;              // [nz][nz]
    double *R = Bt + nz * nz;                 // [nz][nz]
    double *up = R + nz * nz;                 // [nz] B(k,k+1)
    double *dn = up + nz;                     // [nz] B(k,k-1)
    double *dg = dn + nz;                     // [nz] B(k,k)
    double *rsum = dg + nz;                   // [nz]
    double *zm = rsum + nz, *rm = zm + nz, *rw = rm + nz, *kw = rw + nz; // [nz] each
    double *zw = kw + nz;                     // [nz+1]
    double *adv = zw + nz + 1;                // [nz][C][6]: x+, x-, y+, y-, z+, z-
    double *dif = adv + nz * C * 6;           // [nz][4]: x+, x-, y+, y-
    double *au = dif + nz * 4, *ad = au + nz, *ag = ad + nz; // [nz] A's three diagonals
    int *cnt = (int *)(ag + nz);              // [nz][C]
    __shared__ int s_nsteps, s_clamped;
    __shared__ double s_sdt;
    if (t == 0) s_clamped = 0;
    auto cell_of = [&](int k) { return ((int64_t)j * nz + k) * nx + i; };
    auto halo = [&](const float *a, int ii, int k, int jj) {
        return a[((int64_t)(jj + 1) * nz + k) * (int64_t)(nx + 2) + (ii + 1)];
    };
    auto col = [&](const float *a, int k, int nzz) { return a[((int64_t)j * nzz + k) * (int64_t)nx + i]; };

    // phase 0
    for (int x = t; x < nz * C; x += 256) cnt[x] = 0;
    __syncthreads();
    for (int k = wave; k < nz; k += 4) {
        const int c = (int)cell_of(k);
        if (C == 1) { if (lane == 0) cnt[k] = st.n_part[c]; continue; }
        // (several classes: counted by k_class_counts, launched in front of this kernel)
        for (int q = lane; q < C; q += 64) cnt[k * C + q] = st.cls_cnt[(int64_t)c * C + q];
    }
    __syncthreads();

    // phase 1
    for (int x = t; x < nz * C; x += 256) {
        const int k = x / C, cl = x % C;
        const int64_t c = cell_of(k);
        if (f.n_class_part) f.n_class_part[c * C + cl] = cnt[x];
        const double rho_mass = 1.0 / (double)halo(f.alt, i, k, j);
        const double mix_rat = (double)cnt[x] * st.mag[c * C + cl] / rho_mass;
        const float ue = f.u_flux[(((int64_t)cl * ny + j) * nz + k) * (int64_t)(nx + 1) + i + 1];
        const float uw = f.u_flux[(((int64_t)cl * ny + j) * nz + k) * (int64_t)(nx + 1) + i];
        const float vn = f.v_flux[(((int64_t)cl * (ny + 1) + j + 1) * nz + k) * (int64_t)nx + i];
        const float vs = f.v_flux[(((int64_t)cl * (ny + 1) + j) * nz + k) * (int64_t)nx + i];
        const float wu = f.w_flux[(((int64_t)cl * ny + j) * (nz + 1) + k + 1) * (int64_t)nx + i];
        const float wd = f.w_flux[(((int64_t)cl * ny + j) * (nz + 1) + k) * (int64_t)nx + i];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0;
        if (mix_rat > 0.0) {
            if (ue > 0.0f) a0 = (double)ue / mix_rat;
            if (uw < 0.0f) a1 = -1.0 * (double)uw / mix_rat;
            if (vn > 0.0f) a2 = (double)vn / mix_rat;
            if (vs < 0.0f) a3 = -1.0 * (double)vs / mix_rat;
            if (wu > 0.0f) { const double q = (double)wu / mix_rat; a4 = q < .95 ? q : .95; }
            if (wd < 0.0f) { const double q = -1.0 * (double)wd / mix_rat; a5 = q < .95 ? q : .95; }
        }
        const double ps = ((((a3 + a5) + a1) + a0) + a4) + a2; // array element order (:279)
        if (ps > 1.0) {
            const double q = ps + 1.0e-8;
            a0 = a0 / q; a1 = a1 / q; a2 = a2 / q; a3 = a3 / q; a4 = a4 / q; a5 = a5 / q;
        }
        double *a = adv + x * 6;
        a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3; a[4] = a4; a[5] = a5;
    }
    for (int x = t; x < nz * 4; x += 256) {
        const int k = x >> 2, d = x & 3;
        const int di = d == 0 ? 1 : d == 1 ? -1 : 0, dj = d == 2 ? 1 : d == 3 ? -1 : 0;
        const double rho_mass = 1.0 / (double)halo(f.alt, i, k, j);
        const float ksum = halo(f.xkhh, i, k, j) + halo(f.xkhh, i + di, k, j + dj);
        double diff_coef = .5 * (double)ksum;
        if (!(diff_coef > 0.0)) diff_coef = 0.0;
        const double rho_edge = .5 * (rho_mass + (1.0 / (double)halo(f.alt, i + di, k, j + dj)));
        const double h = d < 2 ? f.dx : f.dy;
        dif[x] = (f.dt * diff_coef * rho_edge) / (h * h * rho_mass);
    }

    // phase 2
    bool vertical = true;
    if (!f.all_columns) {
        const int gi = i + f.i_offset + 1, gj = j + f.j_offset + 1;
        vertical = !(gi < 2 || gi > f.global_nx - 1 || gj < 2 || gj > f.global_ny - 1);
    }
    if (vertical) {
        if (wave == 0) {   // stable_timestep: the column's largest exchange coefficient
            float km = -3.0e38f;
            for (int k = lane; k <= nz; k += 64) km = fmaxf(km, col(f.exch_h, k, nz + 1));
            for (int d = 32; d >= 1; d >>= 1) km = fmaxf(km, __shfl_xor(km, d, 64));
            if (lane == 0) {
                const double min_dz = (double)(col(f.z, 1, nz) - col(f.z, 0, nz));
                const double max_k = (double)km > 1.0e-6 ? (double)km : 1.0e-6;
                const double max_time = .1 * (min_dz * min_dz) / max_k;
                double ns = ceil(f.dt / max_time);
                if (!(ns >= 1.0)) ns = 1.0;
                if (ns > TRANS_MAX_SUBSTEPS) { // (flagged: bit 16 of bad; pmcx_transport_probs fails)
                    ns = TRANS_MAX_SUBSTEPS; s_clamped = 1;
                    if (f.clamp_flag) atomicOr(f.clamp_flag, 1);
                }
                s_nsteps = (int)ns;
                s_sdt = f.dt / (double)(int)ns;
            }
        }
        for (int k = t; k <= nz; k += 256) {
            zw[k] = (double)col(f.z_at_w, k, nz + 1);
            if (k < nz) {
                zm[k] = (double)col(f.z, k, nz);
                rm[k] = 1.0 / (double)halo(f.alt, i, k, j);
                double e = k < nz - 1 ? (double)col(f.exch_h, k + 1, nz + 1) : 0.0;
                kw[k] = k < nz - 1 ? (e > 1.0e-6 ? e : 1.0e-6) : 0.0;
            }
        }
        __syncthreads();
        for (int k = t; k < nz - 1; k += 256) {
            const double tmp = zm[k + 1] - zm[k];
            rw[k] = ((zm[k + 1] - zw[k + 1]) / tmp) * rm[k] + ((zw[k + 1] - zm[k]) / tmp) * rm[k + 1];
        }
        __syncthreads();
        const int n_steps = s_nsteps;
        const double sdt = s_sdt;
        for (int k = t; k < nz; k += 256) {
            double d_ = 0.0, u_ = 0.0;
            if (k > 0)
                d_ = (sdt / ((zw[k + 1] - zw[k]) * (zm[k] - zm[k - 1]))) * (rw[k - 1] / rm[k]) * kw[k - 1];
            if (k < nz - 1)
                u_ = (sdt / ((zw[k + 1] - zw[k]) * (zm[k + 1] - zm[k]))) * (rw[k] / rm[k]) * kw[k];
            dn[k] = d_; up[k] = u_;
            dg[k] = k == 0 ? 1.0 - u_ : k == nz - 1 ? 1.0 - d_ : 1.0 - d_ - u_;
        }
        __syncthreads();
        for (int x = t; x < nz * nz; x += 256) {
            const int r = x / nz, q = x % nz;
            Bt[x] = q == r ? dg[r] : q == r + 1 ? up[r] : q == r - 1 ? dn[r] : 0.0;
        }
        __syncthreads();
        for (int it = 2; it <= n_steps; it++) {
            for (int x = t; x < nz * nz; x += 256) {
                const int r = x / nz, q = x % nz;
                double acc = 0.0;
                if (q > 0) acc = acc + Bt[r * nz + q - 1] * up[q - 1];       // B(q-1, q)
                acc = acc + Bt[x] * dg[q];                                    // B(q, q)
                if (q < nz - 1) acc = acc + Bt[r * nz + q + 1] * dn[q + 1];   // B(q+1, q)
                R[x] = acc;
            }
            __syncthreads();
            if (t < nz) {
                double rs = 0.0;
                for (int q = 0; q < nz; q++) rs = rs + R[t * nz + q];
                rsum[t] = rs;
            }
            __syncthreads();
            for (int x = t; x < nz * nz; x += 256) Bt[x] = R[x] / rsum[x / nz];
            __syncthreads();
        }
        if (t == 0 && f.n_substeps) f.n_substeps[(int64_t)j * nx + i] = n_steps;
    } else {
        if (t == 0 && f.n_substeps) f.n_substeps[(int64_t)j * nx + i] = 0;
    }
    __syncthreads();

    // phase 3: per class, R = B* A (or A), then normalize_probs per level and the outputs
    for (int cl = 0; cl < C; cl++) {
        if (vertical) {
            for (int k = t; k < nz; k += 256) {
                const double zp = adv[(k * C + cl) * 6 + 4], zn = adv[(k * C + cl) * 6 + 5];
                au[k] = k < nz - 1 ? zp : 0.0;
                ad[k] = k > 0 ? zn : 0.0;
                ag[k] = k == 0 ? 1.0 - zp : k == nz - 1 ? 1.0 - zn : 1.0 - (zp + zn);
            }
            __syncthreads();
            for (int x = t; x < nz * nz; x += 256) {
                const int r = x / nz, q = x % nz;
                double acc = 0.0;
                if (f.do_vertmix > 0) {
                    if (q > 0) acc = acc + Bt[r * nz + q - 1] * au[q - 1];
                    acc = acc + Bt[x] * ag[q];
                    if (q < nz - 1) acc = acc + Bt[r * nz + q + 1] * ad[q + 1];
                } else {
                    acc = q == r ? ag[r] : q == r + 1 ? au[r] : q == r - 1 ? ad[r] : 0.0;
                }
                R[x] = acc;
            }
        } else {
            for (int x = t; x < nz * nz; x += 256) R[x] = 0.0;
        }
        __syncthreads();
        if (t < nz) {   // normalize_probs of level t (serial sums in the reference's order)
            const int k = t;
            const int64_t c = cell_of(k);
            double a[6], dl[4];
            for (int d = 0; d < 6; d++) a[d] = adv[(k * C + cl) * 6 + d];
            for (int d = 0; d < 4; d++) dl[d] = dif[k * 4 + d];
            double *pv = R + k * nz;
            int flag = 0, neg = 0, big = 0;
            for (int d = 0; d < 6; d++) neg |= !(a[d] >= 0.0);
            for (int d = 0; d < 4; d++) neg |= !(dl[d] >= 0.0);
            for (int q = 0; q < nz; q++) neg |= !(pv[q] >= 0.0);
            if (neg) flag |= 1;
            double ps = ((a[3] + a[1]) + a[0]) + a[2];
            ps = ps + (((dl[3] + dl[1]) + dl[0]) + dl[2]);
            for (int q = 0; q < nz; q++) if (q != k) ps = ps + pv[q];
            if (ps >= 1.0) {
                for (int q = 0; q < nz; q++) pv[q] = pv[q] / ps;
                for (int d = 0; d < 4; d++) { dl[d] = dl[d] / ps; a[d] = a[d] / ps; }
            }
            neg = 0;
            for (int d = 0; d < 6; d++) { neg |= !(a[d] >= 0.0); big |= !(a[d] <= 1.0); }
            for (int d = 0; d < 4; d++) { neg |= !(dl[d] >= 0.0); big |= !(dl[d] <= 1.0); }
            for (int q = 0; q < nz; q++) { neg |= !(pv[q] >= 0.0); big |= !(pv[q] <= 1.0); }
            if (neg) flag |= 2;
            if (big) flag |= 4;
            ps = ((a[3] + a[1]) + a[0]) + a[2];
            ps = ps + (((dl[3] + dl[1]) + dl[0]) + dl[2]);
            for (int q = 0; q < nz; q++) if (q != k) ps = ps + pv[q];
            if (!(ps < 1.0 + 1.0e-8)) flag |= 8;
            double *ph = f.prob_h + (c * C + cl) * 4;   // i-1, i+1, j-1, j+1
            ph[0] = a[1] + dl[1];
            ph[1] = a[0] + dl[0];
            ph[2] = a[3] + dl[3];
            ph[3] = a[2] + dl[2];
            if (s_clamped) flag |= 16;
            if (f.bad) { if (cl == 0) f.bad[c] = flag; else f.bad[c] |= flag; }
        }
        __syncthreads();
        for (int x = t; x < nz * nz; x += 256) {
            const int r = x / nz, q = x % nz;
            f.prob_v[((cell_of(r)) * C + cl) * nz + q] = R[x];
        }
        __syncthreads();
    }
}

// The refill of the lateral boundary cells after the transport
// (interface/wrf_pmc_trans_aero.F90:803-955), one wavefront per cell.
//   k_boundary_copy      outflow: the boundary cell becomes a copy of its inner neighbour
//                        (aero_state_dup_all_particles): same particles, same order, new ids, same
//                        weights
//   k_boundary_resample  inflow: aero_state_resample (:976-1098) against the background -- emptied
//                        and sampled anew, or per class brought to a Poisson target count: short of
//                        it, new particles from the class's modes; over it, a uniformly random
//                        subset of exactly the excess leaves (selection sampling in rank order: the
//                        decisions are serial in the particles, taken lane by lane from uniforms
//                        the wavefront draws 64 at a time)
__global__ __launch_bounds__(64) void k_boundary_copy(DeviceState st, int n_pairs, const int32_t *from,
                                                      const int32_t *to)
{
    const int lane = threadIdx.x, q = blockIdx.x;
    if (q >= n_pairs) return;
    const int cs = from[q], cd = to[q];
    const int64_t sb = (int64_t)cs * st.cap, db = (int64_t)cd * st.cap;
    const int S = st.n_spec, NS = st.n_source, C = st.n_class;
    const uint64_t lt = (1ull << lane) - 1ull;
    const int n_hi = st.n_hi[cs];
    const int64_t nid = st.next_id[cd];
    int base = 0;
    for (int i0 = 0; i0 < n_hi; i0 += 64) {
        const int i = i0 + lane;
        const int wc = i < n_hi ? live_class(st, sb, i) : -1;
        const uint64_t m = __ballot(wc >= 0);
        if (wc >= 0) {
            const int dst = base + __popcll(m & lt);
            for (int sp = 0; sp < S; sp++) st.vol[(db + dst) * S + sp] = spec_vol(st, sb, sp, i);
            for (int sp = 1; sp < NS; sp++) norig(st, db, sp, dst) = norig(st, sb, sp, i);
            st.hot[db + dst] = st.hot[sb + i];
            Cold cv = st.cold[sb + i];
            cv.id = nid + dst;
            st.cold[db + dst] = cv;
            st.wclass[db + dst] = wc;
            comp_copy(st, db + dst, sb + i);
        }
        base += __popcll(m);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lane == 0) {
        st.n_hi[cd] = base; st.n_part[cd] = base;
        st.next_id[cd] = nid + base;
        st.sort_valid[cd] = 0; st.key_valid[cd] = 0; st.n_log[cd] = 0;
    }
    for (int cl = lane; cl < C; cl += 64) st.mag[(int64_t)cd * C + cl] = st.mag[(int64_t)cs * C + cl];
}

__global__ __launch_bounds__(64) void k_boundary_resample(DeviceState st, const int32_t *action,
                                                          AeroDistDev bg, const double *density,
                                                          double create_time, uint64_t seed, uint32_t step,
                                                          int32_t *n_added, int32_t *n_removed)
{
    __shared__ double s_means[64];
    const int lane = threadIdx.x, c = blockIdx.x;
    const int act = action[c];
    if (act == 0) return;
    const int64_t cb = (int64_t)c * st.cap;
    const uint32_t cell_id = cell_gid(st, c);
    const uint64_t lt = (1ull << lane) - 1ull;
    const int C = st.n_class;
    const double dens = density[c];
    int status = 0, na = 0, nr = 0;
    apply_event_log(st, c, lane);
    if (act == 2) { // aero_state_zero, then sampled anew
        nr = st.n_part[c];
        wave_sync();
        if (lane == 0) { st.n_hi[c] = 0; st.n_part[c] = 0; st.sort_valid[c] = 0; st.key_valid[c] = 0; st.n_log[c] = 0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence_block();
        wave_sync();
        int stop_ = -1, cls_cnt = 0; // (a handful of boundary cells: the whole of it here; the cell is empty)
        na = add_dist_sample<true>(st, c, bg, dens, 3u, create_time, seed, step, status, lane, nullptr, 0, stop_, cls_cnt);
    } else {
        for (int cl = 0; cl < C; cl++) {
            const int n_hi = st.n_hi[c];
            int n_before = st.n_part[c];
            if (C > 1) {
                uint32_t cnt = 0;
                for (int i = lane; i < n_hi; i += 64) cnt += (live_class_c(st, cb, i) == cl);
                n_before = (int)wave_sum_u64(cnt);
            }
            double total = 0.0;
            for (int m = 0; m < bg.n_mode; m++) // (a negative num_conc: the cell's background has no such mode)
                if (bg.mode_class[bg.mo(c) + m] == cl && bg.num_conc[(int64_t)c * bg.n_mode + m] >= 0.0)
                    total = total + bg.num_conc[(int64_t)c * bg.n_mode + m] * dens;
            const double mag = st.mag[(int64_t)c * C + cl];
            const double n_parts_new = total / mag;
            uint32_t w[4];
            rng_block(seed, cell_id, step, (uint32_t)cl, SITE_RESAMPLE, 0, w);
            const int n_samp = poisson(n_parts_new, u53(w[0], w[1]));
            const int n_to_add = n_samp - n_before;
            if (n_to_add > 0) {
                wave_sync();
                if (lane < bg.n_mode) {
                    double mean = -1.0;
                    if (bg.mode_class[bg.mo(c) + lane] == cl && bg.num_conc[(int64_t)c * bg.n_mode + lane] >= 0.0) {
                        const double cm = bg.num_conc[(int64_t)c * bg.n_mode + lane];
                        const double share = (dens * cm) / total;
                        const double n_samp_avg = share * (double)n_to_add;
                        const double threshold = (dens * cm / n_samp_avg) * .1;
                        if (mag > threshold) mean = n_samp_avg;
                    }
                    s_means[lane] = mean;
                }
                lds_sync();
                int stop_ = -1, cls_cnt = 0; // (given means: no weight preparation, the counts are not looked at)
                na += add_dist_sample<false>(st, c, bg, dens, 3u, create_time, seed, step, status, lane, s_means, 0, stop_,
                                             cls_cnt);
            } else {
                const int m_rm = -n_to_add < n_before ? -n_to_add : n_before;
                int t0 = 0, k = 0; // class members seen, removed so far (uniform)
                uint32_t rank0 = 0;
                for (int i0 = 0; i0 < n_hi && k < m_rm; i0 += 64) {
                    const int i = i0 + lane;
                    const int wc = i < n_hi ? live_class_c(st, cb, i) : -1;
                    const uint64_t mlive = __ballot(wc >= 0);
                    const uint64_t mcls = __ballot(wc == cl);
                    double u = 0.0;
                    if (wc == cl) {
                        rng_block(seed, cell_id, step, rank0 + (uint32_t)__popcll(mlive & lt), SITE_RESAMPLE,
                                  1u + (uint32_t)cl, w);
                        u = u53(w[0], w[1]);
                    }
                    // the decisions, in rank order (each depends on how many have gone before).  Member t
                    // goes when u (n - t) < m - k, i.e. -- both sides of the comparison being what they
                    // are, an integer on the right -- when k <= m - 1 - floor(u (n - t)): every lane
                    // works out that bound for itself, the serial part is one scalar compare a member
                    const int t_l = t0 + __popcll(mcls & lt);
                    const int a_l = (wc == cl) ? m_rm - 1 - (int)(u * (double)(n_before - t_l)) : -1;
                    uint64_t msel = 0, rest = mcls;
                    while (rest) {
                        const int l = __builtin_ctzll(rest);
                        rest &= rest - 1;
                        if (k <= __builtin_amdgcn_readlane(a_l, l)) { msel |= 1ull << l; k++; }
                    }
                    t0 += __popcll(mcls);
                    if ((msel >> lane) & 1ull) mark_dead(st, cb, i);
                    rank0 += (uint32_t)__popcll(mlive);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                if (lane == 0 && k > 0) { st.n_part[c] = st.n_part[c] - k; st.sort_valid[c] = 0; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __threadfence_block();
                wave_sync();
                nr += k;
            }
        }
        const int hi = st.n_hi[c], live = st.n_part[c];
        if ((hi - live) * st.compact_den > hi) compact_wave(st, c, lane);
    }
    if (lane == 0) {
        st.sort_valid[c] = 0;
        if (n_added) n_added[c] = na;
        if (n_removed) n_removed[c] = nr;
        if (status) atomicOr(&st.status[c], status);
    }
}

// trans_aero_preweight (interface/wrf_pmc_trans_aero.F90:1103-1423): the number concentration
// every (cell, class) is expected to hold after the transport, and a larger weight where that
// is more than twice the ideal particle count.
//   k_preweight_old    a thread per (cell, class): particles per class (k_class_counts) x magnitude
//   k_preweight_new    one thread per (cell, class): compute_fluxes (:1428-1627) turned into a
//                      gather that adds its terms in the order the reference's scatter applies them
//                      to that cell -- sources in (j, k, i) order, the cell's own losses at its place
//                      in that order (j-1, j+1, i-1, i+1, then the levels), what crosses a periodic
//                      edge last (it waits in a ghost cell until the exchange, :1252-1366)
//   k_preweight_scale  one wavefront per cell: aero_state_scale_weight where n_parts_new > 2 ideal
struct PreweightArgs {
    int32_t nx, nz, ny, ideal;
    const double *prob_h, *prob_v, *cell_volume;
    double *old, *expect;   // [n_cells][n_class]
    int32_t *n_removed;     // [n_cells]
    // slab of a domain cut in y across ranks: rows 0 and ny - 1 are ghosts of the neighbours' rows;
    // what those rows send this way arrives as ready terms (the sender's flux x dest_vol /
    // source_vol, [nx nz][n_class] per side; nullptr: nothing) and joins the sum where the
    // whole domain's order has it: in sequence at a cut, after everything else across the
    // domain's periodic edge (lo_edge / hi_edge)
    int32_t slab, lo_edge, hi_edge, j_global0, global_ny;
    const double *gin_lo, *gin_hi;
    double *edge_out;       // [2][nx nz][n_class] what the first and last real rows send out, then (a block
                            // cut in x as well, DeviceState::trans_cut_x) [2][ny nz][n_class] what the first and
                            // last real columns send west and east
    // the same across a cut in x: the neighbours' ready terms, [ny nz][n_class] per side
    int32_t w_edge, e_edge;
    const double *gin_w, *gin_e;
};

__global__ __launch_bounds__(64) void k_preweight_old(DeviceState st, PreweightArgs a)
{
    // (a thread per (cell, class); several classes: counted by k_class_counts, launched in front)
    const int C = st.n_class;
    const int64_t x = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (x >= (int64_t)st.n_cells * C) return;
    const int c = (int)(x / C);
    const int n = C > 1 ? st.cls_cnt[x] : st.n_part[c];
    a.old[x] = (double)n * st.mag[x];
}

// which moves a source cell makes: bit d of the result = towards i-1, i+1, j-1, j+1; bit 4 = levels
__device__ __forceinline__ int preweight_moves(int i, int j, int nx, int ny, int ideal)
{
    // (a periodic direction one cell wide is closed: the cell would be its own neighbour)
    if (ideal) return 16 | (nx > 1 ? 3 : 0) | (ny > 1 ? 12 : 0);
    if (i == 0) return (j != 0 && j != ny - 1) ? 2 : 0;
    if (i == nx - 1) return (j != 0 && j != ny - 1) ? 1 : 0;
    if (j == 0) return 8;
    if (j == ny - 1) return 4;
    return 31;
}

__global__ __launch_bounds__(256) void k_preweight_new(DeviceState st, PreweightArgs a)
{
    const int C = st.n_class, nx = a.nx, nz = a.nz, ny = a.ny;
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= (int64_t)st.n_cells * C) return;
    const int cl = (int)(x % C);
    const int64_t d = x / C;
    const int i = (int)(d % nx), k = (int)((d / nx) % nz), j = (int)(d / ((int64_t)nx * nz));
    if (a.slab && (j == 0 || j == ny - 1)) { a.expect[d * C + cl] = 0.0; return; } // a ghost row
    const bool cut_x = a.slab && st.trans_cut_x;
    if (cut_x && (i == 0 || i == nx - 1)) { a.expect[d * C + cl] = 0.0; return; }   // a ghost column
    auto cell = [&](int ii, int kk, int jj) { return ((int64_t)jj * nz + kk) * nx + ii; };
    // moves of the cell in local row jj, column ii (its place in the whole domain decides at open boundaries)
    const int gny = a.slab ? a.global_ny : ny, gnx = cut_x ? st.trans_gnx : nx;
    auto moves = [&](int ii, int jj) {
        return preweight_moves(cut_x ? ii + st.trans_i0 : ii, a.slab ? jj + a.j_global0 : jj, gnx, gny, a.ideal);
    };
    const double vd = a.cell_volume[d];
    // what source s sends towards this cell through its horizontal move `dir`
    auto from_h = [&](int64_t s, int dir) {
        const double flux = a.prob_h[(s * C + cl) * 4 + dir] * a.old[s * C + cl];
        return flux * (vd / a.cell_volume[s]);
    };
    auto from_v = [&](int64_t s) {
        const double flux = a.prob_v[(s * C + cl) * nz + k] * a.old[s * C + cl];
        return flux * (vd / a.cell_volume[s]);
    };
    const int64_t r = ((int64_t)k * nx + i) * C + cl; // place in a row's table of ready terms
    const int64_t rc = ((int64_t)j * nz + k) * C + cl; // ... and in a column's
    const bool lo_ghost = a.slab && j - 1 == 0, hi_ghost = a.slab && j + 1 == ny - 1;
    const bool w_ghost = cut_x && i - 1 == 0, e_ghost = cut_x && i + 1 == nx - 1;
    double acc = a.old[d * C + cl];
    if (lo_ghost) { if (a.gin_lo && !a.lo_edge) acc = acc + a.gin_lo[r]; }
    else if (j >= 1 && (moves(i, j - 1) & 8)) acc = acc + from_h(cell(i, k, j - 1), 3);
    const bool col_vert = (moves(i, j) & 16) != 0; // (the whole column alike)
    if (col_vert)
        for (int kk = 0; kk < k; kk++) acc = acc + from_v(cell(i, kk, j));
    if (w_ghost) { if (a.gin_w && !a.w_edge) acc = acc + a.gin_w[rc]; }
    else if (i >= 1 && (moves(i - 1, j) & 2)) acc = acc + from_h(cell(i - 1, k, j), 1);
    {   // the cell's own losses, in the order compute_fluxes takes them
        const int mv = moves(i, j);
        const double o = a.old[d * C + cl];
        const double *ph = a.prob_h + (d * C + cl) * 4;
        if (mv & 4) acc = acc - ph[2] * o;
        if (mv & 8) acc = acc - ph[3] * o;
        if (mv & 1) acc = acc - ph[0] * o;
        if (mv & 2) acc = acc - ph[1] * o;
        if (mv & 16)
            for (int kk = 0; kk < nz; kk++)
                if (kk != k) acc = acc - a.prob_v[(d * C + cl) * nz + kk] * o;
    }
    if (e_ghost) { if (a.gin_e && !a.e_edge) acc = acc + a.gin_e[rc]; }
    else if (i + 1 < nx && (moves(i + 1, j) & 1)) acc = acc + from_h(cell(i + 1, k, j), 0);
    if (col_vert)
        for (int kk = k + 1; kk < nz; kk++) acc = acc + from_v(cell(i, kk, j));
    if (hi_ghost) { if (a.gin_hi && !a.hi_edge) acc = acc + a.gin_hi[r]; }
    else if (j + 1 < ny && (moves(i, j + 1) & 4)) acc = acc + from_h(cell(i, k, j + 1), 2);
    if (a.ideal) { // across the periodic edges: the ghost cell's single term, added after everything else
        if (a.slab) {
            if (hi_ghost && a.hi_edge && a.gin_hi) acc = acc + a.gin_hi[r]; // the domain's last row
            if (lo_ghost && a.lo_edge && a.gin_lo) acc = acc + a.gin_lo[r]; // the domain's first row
        } else {
            if (j == ny - 1 && ny > 1) acc = acc + from_h(cell(i, k, 0), 2);
            if (j == 0 && ny > 1) acc = acc + from_h(cell(i, k, ny - 1), 3);
        }
        if (cut_x) {
            if (e_ghost && a.e_edge && a.gin_e) acc = acc + a.gin_e[rc]; // the domain's last column
            if (w_ghost && a.w_edge && a.gin_w) acc = acc + a.gin_w[rc]; // the domain's first column
        } else {
            if (i == nx - 1 && nx > 1) acc = acc + from_h(cell(0, k, j), 0);
            if (i == 0 && nx > 1) acc = acc + from_h(cell(nx - 1, k, j), 1);
        }
    }
    a.expect[d * C + cl] = acc;
}

// slab: what the first real row sends down and the last real row sends up, as ready terms for
// the neighbouring ranks' sums (side 0: towards row 0, side 1: towards row ny - 1)
__global__ __launch_bounds__(256) void k_preweight_edge(DeviceState st, PreweightArgs a)
{
    const int C = st.n_class, nx = a.nx, nz = a.nz, ny = a.ny;
    const int64_t row = (int64_t)nx * nz * C, col = (int64_t)ny * nz * C;
    const bool cut_x = st.trans_cut_x != 0;
    const int gnx = cut_x ? st.trans_gnx : nx, i0 = cut_x ? st.trans_i0 : 0;
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= 2 * row + (cut_x ? 2 * col : 0)) return;
    int i, k, j, ig, jg, dir, bit, cl;
    if (x < 2 * row) { // side 0: towards row 0, side 1: towards row ny - 1
        const int side = (int)(x / row);
        const int64_t r = x % row;
        cl = (int)(r % C); i = (int)((r / C) % nx); k = (int)(r / ((int64_t)C * nx));
        j = side == 0 ? 1 : ny - 2; jg = side == 0 ? 0 : ny - 1; ig = i;
        dir = side == 0 ? 2 : 3; bit = side == 0 ? 4 : 8;
    } else {           // side 2: towards column 0, side 3: towards column nx - 1
        const int64_t y = x - 2 * row;
        const int side = (int)(y / col);
        const int64_t r = y % col;
        cl = (int)(r % C); k = (int)((r / C) % nz); j = (int)(r / ((int64_t)C * nz));
        i = side == 0 ? 1 : nx - 2; ig = side == 0 ? 0 : nx - 1; jg = j;
        dir = side == 0 ? 0 : 1; bit = side == 0 ? 1 : 2;
    }
    const int64_t s = ((int64_t)j * nz + k) * nx + i, g = ((int64_t)jg * nz + k) * nx + ig;
    double term = 0.0;
    if (preweight_moves(i + i0, j + a.j_global0, gnx, a.global_ny, a.ideal) & bit) {
        const double flux = a.prob_h[(s * C + cl) * 4 + dir] * a.old[s * C + cl];
        term = flux * (a.cell_volume[g] / a.cell_volume[s]);
    }
    a.edge_out[x] = term;
}

__global__ __launch_bounds__(64) void k_preweight_scale(DeviceState st, PreweightArgs a, uint64_t seed,
                                                        uint32_t step)
{
    const int lane = threadIdx.x, c = blockIdx.x, C = st.n_class;
    const double *mag = st.mag + (int64_t)c * C;
    uint32_t removed = 0;
    int status = 0;
    for (int cl = 0; cl < C; cl++) {
        const double n_parts_new = a.expect[(int64_t)c * C + cl] / ld_f64(&mag[cl]);
        const double ideal_n = st.n_ideal[(int64_t)c * C + cl];
        if (!(n_parts_new > 2.0 * ideal_n)) continue; // (uniform over the wavefront)
        // aero_state_scale_weight(weight_ratio > 1, .true., .true.) (:1390-1393); a cell past its
        // tombstone threshold afterwards: the k_compact that follows
        removed += (uint32_t)scale_weight_wave<false>(st, c, cl, n_parts_new / ideal_n, 1, 1, seed, step, (uint32_t)cl, lane,
                                               status);
    }
    if (lane == 0 && a.n_removed) a.n_removed[c] = (int32_t)removed;
}

// Arrivals are appended behind n_hi.  A cell that would run out of slots while it has holes
// closes them now (stable, as always) and renames the slots its own outbox names; this is the
// only moment it can: the outboxes exist, nobody has read them yet.
__global__ __launch_bounds__(256) void k_trans_room(DeviceState st, TransGrid tg)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // four cells to a workgroup; the few that need room are compacted by the whole workgroup, one
    // after the other (compact_group: the chain of trips is the kernel's duration)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_chunk = (st.cap + 63) / 64 + 1;
    uint64_t *cmask = (uint64_t *)smem;
    uint32_t *cbase = (uint32_t *)(cmask + (size_t)GW * n_chunk);
    int *s_cnt = (int *)(cbase + (size_t)GW * n_chunk);
    int *s_need = s_cnt + GW * GU + 2;
    {
        const int c = blockIdx.x * GW + wave;
        bool need = false;
        if (c < st.n_cells) {
            const int n_hi = st.n_hi[c], dead = n_hi - st.n_part[c];
            need = dead > 0 && n_hi + tg.n_exp[c] > st.cap;
        }
        if (lane == 0) s_need[wave] = need;
    }
    __syncthreads();
    for (int w = 0; w < GW; w++) {
        if (!s_need[w]) continue; // (the same for every thread)
        const int c = blockIdx.x * GW + w;
        compact_group(st, c, threadIdx.x, cbase, cmask, true, s_cnt);
        int2 *box = tg.outbox + (int64_t)c * st.cap;
        const int n_out = tg.n_out[c];
        for (int e = threadIdx.x; e < n_out; e += 64 * GW) {
            const int2 en = box[e];
            const uint32_t sl = (uint32_t)en.x & 0xFFFFu, k = sl >> 6;
            const uint32_t nw = cbase[k] + (uint32_t)__popcll(cmask[k] & ((1ull << (sl & 63)) - 1ull));
            box[e] = make_int2((int)((uint32_t)en.x & 0xFFFF0000u) | (int)nw, en.y);
        }
        __syncthreads(); // (the slot map is rewritten by the next cell)
    }
}

__global__ __launch_bounds__(64) void k_trans_gather(DeviceState st, TransGrid tg)
{
    // The arrivals of a cell are the entries addressed to it in the outboxes of its sources, in
    // the fixed order of the sources and, within one, of its outbox.  They are few among what
    // is scanned (a source sends to half a dozen cells), so the scan only collects them -- the
    // sources' counts requested by a lane each at once, the outboxes walked as ONE sequence,
    // the next chunk of entries requested before the current one is looked at -- into a small
    // queue in LDS, and the records are moved 64 arrivals at a time from the queue: dense
    // lanes and a memory round trip per 64 arrivals instead of two per 64 scanned entries.
    __shared__ int32_t s_qsc[128], s_qx[128]; // queue: source cell, slot | copies << 16
    const int lane = threadIdx.x, c = tg.c_first + blockIdx.x;
    const int nx = tg.nx, nz = tg.nz, ny = tg.ny;
    const int i = c % nx, k = (c / nx) % nz, j = c / (nx * nz);
    if (ghost_col(st, i, nx) || (st.trans_lateral && (lateral_col(st, i, nx) || lateral_row(st, j, ny)))) {
        if (lane == 0) tg.n_arr[c] = 0; // an open lateral boundary cell takes nothing in (:775-801); nor does a ghost
        return;
    }
    const int64_t cb = (int64_t)c * st.cap;
    const int S = st.n_spec, NS = st.n_source;
    const int n_hi0 = st.n_hi[c];
    int at = n_hi0;                 // next free slot of this cell
    int64_t nid = st.next_id[c];
    int status = 0;
    int queued = 0;                 // entries in the queue (uniform)

    // move the first `take` queued arrivals (records, keys, ids), in queue order
    auto emit = [&](int take) {
        const int sc = lane < take ? s_qsc[lane] : 0, x = lane < take ? s_qx[lane] : 0;
        const int copies = lane < take ? ((x >> 16) & TRANS_COPIES_MASK) : 0, from = x & 0xFFFF;
        const bool stays = x < 0; // (an emitter's copies: the source keeps its particle, every copy a new id)
        const bool keep = (x & TRANS_KEEP) != 0; // (a mover that also emits: k_trans_remove leaves its tombstone)
        const int64_t sb = (int64_t)sc * st.cap;
        // slots and new ids of this entry's copies: prefix sums over the chunk
        const uint32_t inc = wave_incl_scan_u32((uint32_t)copies, lane);
        const uint32_t first = (uint32_t)at + inc - (uint32_t)copies;
        const uint32_t n_new = copies > 0 ? (uint32_t)(stays ? copies : copies - 1) : 0u;
        const uint32_t inc_new = wave_incl_scan_u32(n_new, lane);
        const int64_t id0 = nid + (int64_t)(inc_new - n_new);
        if (copies > 0) {
            // the record is read once (one request per plane), then written per copy; with
            // one species the species plane is not kept current (see spec_vol): not moved
            const Hot hv = st.hot[sb + from];
            const Cold cv = st.cold[sb + from];
            const int32_t wc = st.n_class > 1 ? st.wclass[sb + from] : 0; // one class: all zero
            const uint16_t kd = key_of(st, hv.vtot, wc); // (from the record: the source's keys may not hold)
            // the mover's tombstone, into the line just read (k_trans_remove leaves these alone:
            // a store to a line nobody has read costs a read-modify-write in memory)
            if (!stays && !keep) st.hot[sb + from].vtot = -1.0;
            for (int g = 0; g < copies; g++) {
                const int dst = (int)first + g;
                if (dst >= st.cap) { status |= ST_CAP_OVERFLOW; break; }
                for (int sp = 1; sp < NS; sp++)
                    norig(st, cb, sp, dst) = norig(st, sb, sp, from);
                st.hot[cb + dst] = hv;
                st.gkey[cb + dst] = kd;
                if (st.n_class > 1) st.wclass[cb + dst] = wc;
                Cold c2 = cv;
                if (stays || g != copies - 1) c2.id = id0 + g;
                st.cold[cb + dst] = c2;
                comp_copy(st, cb + dst, sb + from);
            }
        }
        if (S > 1) {
            // the species volumes of the chunk's arrivals, lanes over (arrival, species): n_spec x 8 contiguous
            // bytes per record, four requests a lane in flight (a lane copying its own arrival's species one
            // after the other was twenty dependent round trips a chunk at the reference's 20 species)
            const int64_t srow = sb + from;
            const int fits = copies > 0 ? max(0, min(copies, st.cap - (int)first)) : 0;
            const int s_lo = (int)(uint32_t)srow, s_hi = (int)(uint32_t)((uint64_t)srow >> 32);
            for (int it0 = 0; it0 < 64 * S; it0 += 256) {
                double v[4];
                int64_t d0[4];
                int n4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int it = it0 + u * 64 + lane;
                    const int e = min(it / S, 63), sp = it - (it / S) * S;
                    const int64_t src = (int64_t)(((uint64_t)(uint32_t)__shfl(s_hi, e, 64) << 32) | (uint32_t)__shfl(s_lo, e, 64));
                    const int df = __shfl((int)first, e, 64), nc = __shfl(fits, e, 64);
                    n4[u] = it < 64 * S ? nc : 0;
                    d0[u] = (cb + df) * S + sp;
                    v[u] = n4[u] > 0 ? st.vol[src * S + sp] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    for (int g = 0; g < n4[u]; g++) st.vol[d0[u] + (int64_t)g * S] = v[u];
            }
        }
        at += (int)__shfl(inc, 63, 64);
        nid += (int64_t)__shfl(inc_new, 63, 64);
        // what is left moves to the front of the queue
        lds_sync();
        const int rest = queued - take;
        int rsc = 0, rx = 0;
        if (lane < rest) { rsc = s_qsc[take + lane]; rx = s_qx[take + lane]; }
        lds_sync();
        if (lane < rest) { s_qsc[lane] = rsc; s_qx[lane] = rx; }
        queued = rest;
        lds_sync();
    };

    // sources in the fixed order: east, west, north, south neighbour, then the column -- a lane
    // each, 64 at a time
    for (int q0 = 0; q0 < 4 + nz; q0 += 64) {
        const int q = q0 + lane;
        int sc = -1;
        if (q < 4) {
            const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
            int ii = i + di[q], jj = j + dj[q];
            bool ok = true;
            if (ii < 0 || ii >= nx) { if (!tg.periodic_x) ok = false; else ii = (ii + nx) % nx; }
            if (jj < 0 || jj >= ny) { if (!tg.periodic_y) ok = false; else jj = (jj + ny) % ny; }
            if (ok) sc = (jj * nz + k) * nx + ii;
            // under periodic wrap with nx or ny <= 2 a neighbour repeats (or is the cell itself)
            for (int q2 = 0; q2 < q && sc >= 0; q2++) {
                int i2 = i + di[q2], j2 = j + dj[q2];
                bool ok2 = true;
                if (i2 < 0 || i2 >= nx) { if (!tg.periodic_x) ok2 = false; else i2 = (i2 + nx) % nx; }
                if (j2 < 0 || j2 >= ny) { if (!tg.periodic_y) ok2 = false; else j2 = (j2 + ny) % ny; }
                if (ok2 && (j2 * nz + k) * nx + i2 == sc) sc = -1;
            }
            if (sc == c) sc = -1;
        } else if (q < 4 + nz) {
            const int kk = q - 4;
            if (kk != k) sc = (j * nz + kk) * nx + i;
        }
        // does that source send anything this way?  (this cell as seen from the source)
        int n_src = 0;
        if (sc >= 0) {
            const int need = q < 4 ? (q ^ 1) : 4 + k;
            const uint64_t dm = tg.dest_mask[sc];
            const int no = tg.n_out[sc];
            if (need >= 64 || ((dm >> need) & 1ull)) n_src = no;
        }
        const uint32_t incl = wave_incl_scan_u32((uint32_t)n_src, lane);
        const int start = (int)(incl - (uint32_t)n_src);
        const int total = (int)__shfl(incl, 63, 64);
        const uint64_t act = __ballot(n_src > 0);
        // entry f of the sources' outboxes laid end to end
        auto fetch = [&](int f, int &fsc) -> int2 {
            int2 en = make_int2(0, -1);
            int sq = 0, e = 0;
            for (uint64_t m = act; m; m &= m - 1) { // (a handful of sources: every lane walks them)
                const int sl = __builtin_ctzll(m);
                const int st_s = __builtin_amdgcn_readlane(start, sl);
                const int sc_s = __builtin_amdgcn_readlane(sc, sl);
                if (f >= st_s) { sq = sc_s; e = f - st_s; }
            }
            fsc = sq;
            if (f < total) en = tg.outbox[(int64_t)sq * st.cap + e];
            return en;
        };
        int nsc = 0;
        int2 nen = make_int2(0, -1);
        if (total > 0) nen = fetch(lane, nsc);
        for (int f0 = 0; f0 < total; f0 += 64) {
            const int2 en = nen;
            const int esc = nsc;
            if (f0 + 64 < total) nen = fetch(f0 + 64 + lane, nsc); // requested before this chunk's work
            const bool mine = en.y == c && ((en.x >> 16) & TRANS_COPIES_MASK) > 0;
            const uint64_t mm = __ballot(mine);
            if (mine) {
                const int pos = queued + __popcll(mm & ((1ull << lane) - 1ull));
                s_qsc[pos] = esc; s_qx[pos] = en.x;
            }
            queued += __popcll(mm);
            lds_sync();
            if (queued >= 64) emit(64);
        }
    }
    if (queued > 0) emit(queued);
    if (__ballot(status != 0)) status = ST_CAP_OVERFLOW;
    if (at > st.cap) at = st.cap;
    if (lane == 0) {
        st.n_hi[c] = at;
        st.n_part[c] = st.n_part[c] + (at - n_hi0);
        st.next_id[c] = nid;
        tg.n_arr[c] = at - n_hi0;
        if (status) atomicOr(&st.status[c], status);
    }
}

__global__ __launch_bounds__(64) void k_trans_remove(DeviceState st, TransGrid tg)
{
    const int lane = threadIdx.x, c = tg.c_first + blockIdx.x;
    const int64_t cb = (int64_t)c * st.cap;
    const int n_out = tg.n_out[c];
    const int2 *box = tg.outbox + cb;
    // Tombstones of the movers no gather has left one for: those that arrive nowhere (no copy
    // survives the change of weight), go to a cell outside the rows gathered here (a ghost
    // row: packed for the neighbouring rank) or to an open lateral boundary cell.  The sort
    // keys were seen to by k_trans_decide.
    const int nx = tg.nx, nz = tg.nz, ny = tg.ny;
    if (ghost_col(st, c % nx, nx)) return; // (a ghost column's cells are emptied by k_halo_clear)
    int n_movers = 0;
    for (int e0 = 0; e0 < n_out; e0 += 64) {
        const int e = e0 + lane;
        const int2 en = e < n_out ? box[e] : make_int2(TRANS_STAYS, -1);
        const bool stays = en.x < 0; // (an emitter's copy, law 1: its particle is not going anywhere)
        n_movers += __popcll(__ballot(!stays));
        if (stays) continue;
        const int dest = en.y;
        bool gathered = (en.x & TRANS_KEEP) == 0 &&  // (a mover that also emits: its gather left the record alone)
                        ((en.x >> 16) & TRANS_COPIES_MASK) > 0 && dest >= tg.c_first && dest < tg.c_first + (int)gridDim.x &&
                        !ghost_col(st, dest % nx, nx); // (to a ghost column: packed for the neighbouring rank)
        if (gathered && st.trans_lateral) {
            const int di = dest % nx, dj = dest / (nx * nz);
            if (lateral_col(st, di, nx) || lateral_row(st, dj, ny)) gathered = false;
        }
        if (!gathered) st_f64(&st.hot[cb + (en.x & 0xFFFF)].vtot, -1.0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    const int n_live = st.n_part[c] - n_movers;
    if (lane == 0) {
        st.n_part[c] = n_live;
        tg.n_out[c] = n_movers; // (what the caller reads as n_moved: the particles that left, not the entries)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    wave_sync();
    // (a cell past its tombstone threshold is compacted by the k_compact that follows: a workgroup
    // per such cell, not this cell's one wavefront)
}

// ------------------------------------------------------- transport across ranks
//
// The domain cut in y (slabs) or in x and y (the reference's two directions,
// wrf_pmc_trans_aero.F90:148-273): a rank's block carries one ghost row below its first real row
// and one above its last, and -- cut in x as well -- one ghost column west and east.  After
// k_trans_decide the movers of a boundary strip (a row or a column: Strip) that go to the
// neighbouring rank are packed (source cell by source cell, in outbox order) and shipped; the
// receiver lays them out in its ghost strip exactly as the sender's boundary strip looked to a
// destination -- particles in slots 0.., an outbox that names them -- so that k_trans_gather,
// unchanged, pulls them in the same place of its fixed source order as it would inside one
// block.  A mover goes to one of its four neighbours or a level of its column: corners carry
// nothing.  Record: header {src, dest (cell index within its strip), copies, wclass, index within
// the source cell, pad} | vtot, mass | lct, gct, id, n_orig0 | vol[n_spec] | n_orig[1..] (padded
// to 8 bytes) | component list (n_comp_max > 0).

__host__ __device__ inline size_t halo_rec_bytes(int n_spec, int n_source, int n_comp_max)
{
    const size_t extra = n_source > 1 ? (size_t)(n_source - 1) * 4 : 0;
    const size_t comp = n_comp_max ? 8 * (size_t)(1 + n_comp_max) : 0; // source word, create times
    return 24 + 16 + 32 + 8 * (size_t)n_spec + ((extra + 7) & ~(size_t)7) + comp;
}

// A strip is a row of the block (cells base + r, r < nx nz) or, for a block cut in x as well, a column
// (cells i0 + r nx, r < ny nz).  Place of a cell in a strip, -1 when it is not in it:
__device__ __forceinline__ int strip_index(const Strip &t, int cell)
{
    const int d = cell - t.base;
    if (d < 0 || d % t.stride != 0) return -1;
    const int r = d / t.stride;
    return r < t.n ? r : -1;
}

// entries of the cells of strip src that go to strip dst: counts per source cell
__global__ __launch_bounds__(64) void k_halo_count(DeviceState st, TransGrid tg, Strip src, Strip dst,
                                                   int32_t *counts)
{
    const int lane = threadIdx.x, r = blockIdx.x; // r: cell index within the strip
    const int c = src.base + r * src.stride;
    const int2 *box = tg.outbox + (int64_t)c * st.cap;
    const int n_out = tg.n_out[c];
    int cnt = 0;
    for (int e0 = 0; e0 < n_out; e0 += 64) {
        const int e = e0 + lane;
        const bool mine = e < n_out && strip_index(dst, box[e].y) >= 0;
        cnt += __popcll(__ballot(mine));
    }
    if (lane == 0) counts[r] = cnt;
}

// exclusive scan of n counts by one wavefront; offsets[n] = total
__global__ __launch_bounds__(64) void k_halo_scan(const int32_t *counts, int32_t *offsets, int n)
{
    const int lane = threadIdx.x;
    uint32_t base = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const uint32_t x = j < n ? (uint32_t)counts[j] : 0u;
        const uint32_t inc = wave_incl_scan_u32(x, lane);
        if (j < n) offsets[j] = (int32_t)(base + inc - x);
        base += __shfl(inc, 63, 64);
    }
    if (lane == 0) offsets[n] = (int32_t)base;
}

__global__ __launch_bounds__(64) void k_halo_pack(DeviceState st, TransGrid tg, Strip src, Strip dst,
                                                  const int32_t *offsets, unsigned char *recs, int rec_bytes)
{
    const int lane = threadIdx.x, r = blockIdx.x;
    const int c = src.base + r * src.stride;
    const int64_t cb = (int64_t)c * st.cap;
    const int2 *box = tg.outbox + cb;
    const int n_out = tg.n_out[c];
    const int S = st.n_spec, NS = st.n_source;
    const uint64_t lt = (1ull << lane) - 1ull;
    int at = 0;
    for (int e0 = 0; e0 < n_out; e0 += 64) {
        const int e = e0 + lane;
        int2 en = make_int2(0, -1);
        if (e < n_out) en = box[e];
        const int didx = e < n_out ? strip_index(dst, en.y) : -1;
        const bool mine = didx >= 0;
        const uint64_t m = __ballot(mine);
        if (mine) {
            const int idx = at + __popcll(m & lt);
            unsigned char *rec = recs + (size_t)(offsets[r] + idx) * rec_bytes;
            const int from = en.x & 0xFFFF;
            int32_t *h = (int32_t *)rec;
            h[0] = r; h[1] = didx; h[2] = (en.x >> 16) & TRANS_COPIES_MASK;
            h[3] = st.wclass[cb + from]; h[4] = idx; h[5] = en.x < 0 ? 1 : 0; // (the source particle stays: law 1)
            double *d = (double *)(rec + 24);
            d[0] = st.hot[cb + from].vtot; d[1] = st.hot[cb + from].mass;
            d[2] = st.hot[cb + from].lct; d[3] = st.hot[cb + from].gct;
            ((int64_t *)d)[4] = st.cold[cb + from].id;
            ((int32_t *)(d + 5))[0] = st.cold[cb + from].n_orig0; ((int32_t *)(d + 5))[1] = 0;
            for (int sp = 0; sp < S; sp++) d[6 + sp] = spec_vol(st, cb, sp, from);
            int32_t *no = (int32_t *)(d + 6 + S);
            for (int sp = 1; sp < NS; sp++) no[sp - 1] = norig(st, cb, sp, from);
            if (st.n_comp_max) {
                double *cm = (double *)(rec + rec_bytes - 8 * (1 + st.n_comp_max));
                ((uint64_t *)cm)[0] = st.comp_src[cb + from];
                for (int q = 0; q < st.n_comp_max; q++) cm[1 + q] = st.comp_time[(cb + from) * st.n_comp_max + q];
            }
        }
        at += __popcll(m);
    }
}

// the receiving side: ghost cell r of row ghost_row gets its records' particles in slots 0..,
// an outbox naming them with destinations in row real_row, and tells those cells what is coming
__global__ __launch_bounds__(64) void k_halo_unpack(DeviceState st, TransGrid tg, Strip ghost, Strip real,
                                                    const int32_t *counts, const int32_t *offsets,
                                                    const unsigned char *recs, int rec_bytes,
                                                    uint64_t mask_bit, int32_t *bad)
{
    const int lane = threadIdx.x, r = blockIdx.x;
    const int c = ghost.base + r * ghost.stride;
    const int64_t cb = (int64_t)c * st.cap;
    int2 *box = tg.outbox + cb;
    const int S = st.n_spec, NS = st.n_source;
    int cnt = counts[r]; // (the host has checked 0 <= cnt <= cap and that the records fit the buffer)
    cnt = cnt < 0 ? 0 : (cnt > st.cap ? st.cap : cnt);
    for (int e = lane; e < cnt; e += 64) {
        const unsigned char *rec = recs + (size_t)(offsets[r] + e) * rec_bytes;
        const int32_t *h = (const int32_t *)rec;
        const int slot = e; // records of a cell arrive in outbox order (h[4] repeats it)
        int copies = h[2];
        // the buffer was written by another process: what indexes this block is checked; a
        // record that fails becomes an entry without copies (dropped) and is counted
        const bool ok = h[4] == e && (unsigned)h[1] < (unsigned)real.n &&
                        (unsigned)copies < 32768u && (unsigned)h[3] < (unsigned)st.n_class;
        if (!ok) { atomicAdd(bad, 1); copies = 0; }
        const int dest = real.base + (ok ? h[1] : 0) * real.stride;
        const double *d = (const double *)(rec + 24);
        st.hot[cb + slot].vtot = d[0]; st.hot[cb + slot].mass = d[1];
        st.hot[cb + slot].lct = d[2]; st.hot[cb + slot].gct = d[3];
        st.cold[cb + slot].id = ((const int64_t *)d)[4];
        st.cold[cb + slot].n_orig0 = ((const int32_t *)(d + 5))[0];
        st.wclass[cb + slot] = ok ? h[3] : 0;
        for (int sp = 0; sp < S; sp++) st.vol[(cb + slot) * S + sp] = d[6 + sp];
        const int32_t *no = (const int32_t *)(d + 6 + S);
        for (int sp = 1; sp < NS; sp++) norig(st, cb, sp, slot) = no[sp - 1];
        if (st.n_comp_max) {
            const double *cm = (const double *)(rec + rec_bytes - 8 * (1 + st.n_comp_max));
            st.comp_src[cb + slot] = ((const uint64_t *)cm)[0];
            for (int q = 0; q < st.n_comp_max; q++) st.comp_time[(cb + slot) * st.n_comp_max + q] = cm[1 + q];
        }
        box[slot] = make_int2(slot | (copies << 16) | ((ok && h[5] == 1) ? TRANS_STAYS : 0), dest);
        if (copies > 0) atomicAdd(&tg.n_exp[dest], copies);
    }
    if (lane == 0) {
        tg.n_out[c] = cnt;
        tg.dest_mask[c] = cnt ? mask_bit : 0ull;
        st.n_hi[c] = cnt;
        st.n_part[c] = cnt;
    }
}

// ghost rows are scratch: empty again once the step is over
__global__ void k_halo_clear(DeviceState st, TransGrid tg, Strip a, Strip b)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n + b.n) return;
    const int c = j < a.n ? a.base + j * a.stride : b.base + (j - a.n) * b.stride;
    st.n_hi[c] = 0; st.n_part[c] = 0; tg.n_out[c] = 0; tg.dest_mask[c] = 0; st.sort_valid[c] = 0;
    st.key_valid[c] = 0;
}

// ------------------------------------------------------------- condensation
//
// EXTENSION (see pmcx_device_math.h): water uptake of every live particle over
// del_t at the cell's (T, p, RH): n_sub forward-Euler sub-steps of the growth
// law, a sub-step that would pass the kappa-Koehler equilibrium diameter lands
// on it.  One thread per particle; reads the species volumes, writes the water
// volume and the caches.  fp64-VALU-bound (about 16 growth-law evaluations
// per particle), not HBM-bound.

__global__ __launch_bounds__(256) void k_condense(DeviceState st, const double *kappa,
                                                  const double *rel_humid, double del_t, int n_sub)
{
    const int c = blockIdx.y;
    const int64_t cb = (int64_t)c * st.cap;
    const int n = st.n_hi[c];
    const int S = st.n_spec, iw = st.i_water;
    double rh = rel_humid[c];
    if (rh > 0.999) rh = 0.999;
    const CondEnv e = cond_env_derive(st.temp[c], st.pressure[c], rh);
    const double h = del_t / (double)n_sub;
    // A particle's species are contiguous: one pass over its record gives the dry volume, the
    // kappa-weighted volume and the dry mass.  When water is the last species (the reference's
    // table, aero_data.dat:21) the new cached totals are those sums plus the water term -- the
    // same additions in the same order as a sum over all species -- so nothing is read twice.
    const bool water_last = (iw == S - 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (live_class(st, cb, i) < 0) continue; // tombstone
        const double *rec = st.vol + (cb + i) * S;
        double vdry = 0.0, kv = 0.0, mdry = 0.0, v_water = 0.0;
        if ((S & 1) == 0) { // 16-byte loads: two species each
            const double2 *r2 = (const double2 *)rec;
            for (int s = 0; s < S; s += 2) {
                const double2 v = r2[s >> 1];
                if (s == iw) v_water = v.x;
                else { vdry = vdry + v.x; kv = kv + kappa[s] * v.x; mdry = mdry + st.density[s] * v.x; }
                if (s + 1 == iw) v_water = v.y;
                else { vdry = vdry + v.y; kv = kv + kappa[s + 1] * v.y; mdry = mdry + st.density[s + 1] * v.y; }
            }
        } else {
            for (int s = 0; s < S; s++) {
                const double v = rec[s];
                if (s == iw) v_water = v;
                else { vdry = vdry + v; kv = kv + kappa[s] * v; mdry = mdry + st.density[s] * v; }
            }
        }
        if (!(vdry > 0.0) || !(kv > 0.0)) continue;
        const double kap = kv / vdry;
        const double dd3 = vdry * (6.0 / PI);
        const double dd = det_cbrt(dd3);
        const double d3 = (vdry + v_water) * (6.0 / PI);
        double d = det_cbrt(d3);
        const double d_min = dd * (1.0 + 1e-9);
        if (d < d_min) d = d_min;
        const double d_eq = cond_equilibrium(dd, dd3, kap, d_min, e);
        for (int sub = 0; sub < n_sub; sub++) {
            const bool below = d < d_eq;
            double dn = d + h * cond_rate(d, dd3, kap, e);
            if (below ? !(dn < d_eq) : !(dn > d_eq)) dn = d_eq;
            d = dn;
        }
        double vw = d * d * d * (PI / 6.0) - vdry;
        if (vw < 0.0) vw = 0.0;
        st.vol[(cb + i) * S + iw] = vw;
        double vt, ms;
        if (water_last) {
            vt = vdry + vw;
            ms = mdry + st.density[iw] * vw;
        } else {
            vt = 0.0; ms = 0.0;
            for (int s = 0; s < S; s++) {
                const double v = (s == iw) ? vw : rec[s];
                vt = vt + v;
                ms = ms + st.density[s] * v;
            }
        }
        st.hot[cb + i].vtot = vt;
        st.hot[cb + i].mass = ms;
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
    const uint32_t cell_id = cell_gid(st, c);
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
            st.vol[(cb + i) * st.n_spec + s] = vs;
            vt = vt + vs;
            ms = ms + st.density[s] * vs;
        }
        for (int s = 0; s < st.n_source; s++)
            norig(st, cb, s, i) = (s == g.mode_source[mode]) ? 1 : 0;
        st.hot[cb + i].vtot = vt;
        st.hot[cb + i].mass = ms;
        st.wclass[cb + i] = g.mode_class[mode];
        st.cold[cb + i].id = ((int64_t)cell_id << 32) + i;
        st.hot[cb + i].lct = create_time;
        st.hot[cb + i].gct = create_time;
        comp_single(st, cb + i, g.mode_source[mode], create_time);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st.n_part[c] = n; st.n_hi[c] = n; }
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
        const PartProps qa = props_for(ktype, a[i], b[i], env);
        const PartProps qb = props_for(ktype, c3[i], c4[i], env);
        r = kernel_pair(ktype, a[i], c3[i], qa, qb, kp);
        break;
    }
    case 8: // dry-deposition velocity: diameter, density, ustar, aer_res_a; pressure = 1/rho_air;
            // ktype 0: grass (alpha 1.2, A 2 mm), 1: water (alpha 100, A = 9.999: the A > 1 branch)
        r = dep_velocity(a[i], b[i], temp, pressure, c3[i], c4[i], ktype ? 100.0 : 1.2,
                         ktype ? 9.999 : 0.002);
        break;
    case 9: { // the single-precision screen of the Brownian kernel; negative: the pair is
              // outside the range where the screen is trusted (it takes the fp64 path)
        const EnvDerived env = env_derive(temp, pressure);
        EnvF32 ef;
        ef.gasfreepath = (float)env.gasfreepath; ef.inv_gfp = (float)env.inv_gfp;
        ef.c_diff = (float)env.c_diff; ef.c_speed = (float)env.c_speed;
        ef.c_inv_speed = (float)env.c_inv_speed;
        bool ok;
        const float k32 = brown_screen_f32(a[i], b[i], c3[i], c4[i], ef, ok);
        r = (ok && k32 > 1.0e-30f) ? (double)k32 : -1.0;
        if (ktype == 1000) r = (double)k32; // (test hook: the estimate whether trusted or not)
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

// particles per class of every cell into st.cls_cnt (several classes)
static void launch_class_counts(const DeviceState &st, hipStream_t s, bool reuse = false)
{
    if (st.n_class <= 1 || !st.cls_cnt) return;
    if (reuse && st.cls_fresh) return; // (counted by the call in front of this one, nothing since)
    const int n_wg = (st.n_cells + 3) / 4;
    hipLaunchKernelGGL(k_class_counts, dim3(n_wg < 32768 ? n_wg : 32768), dim3(256), 4 * 8 * sizeof(int) * (size_t)st.n_class, s,
                       st, st.cls_cnt);
}

static size_t bin_sort_lds_for(const DeviceState &st, int nw, bool stage)
{
    const size_t B = st.n_bin, G = (size_t)st.n_bin * st.n_class;
    const size_t o_tab = (sizeof(double) * (B + 1) + 15) & ~(size_t)15;
    const size_t o_key = (o_tab + (sizeof(uint64_t) * nw + sizeof(uint32_t) * (nw + 1)) * G + 15) & ~(size_t)15;
    // (without the copy: two halves of 512 keys per wavefront)
    return o_key + (stage ? sizeof(uint16_t) * (((size_t)st.cap + 7) & ~(size_t)7) : sizeof(uint16_t) * 1024 * (size_t)nw);
}
// The keys' whole copy in LDS while five workgroups share a CU with it (30 KB: the headline's cap = 10 048
// with one class, 9.85 ms for 3*10^9 slots against 10.0 without the copy), or where the keys cannot be
// taken eight to a request (cap not a multiple of 8); else 512 at a time through the wavefront's rolling
// buffer -- the coupled step's cap = 22 016 with one class: 6.4 -> 5.7 ms per launch, a chemistry step
// 13.7 -> 12.8 ms (scripts/gpu_ab_sort.sh); ten classes: 15.7 -> 9.6 ms
static bool bin_sort_stage(const DeviceState &st)
{
    const size_t with_copy = bin_sort_lds_for(st, 4, true);
    return with_copy <= (size_t)30 * 1024 || ((st.cap & 7) != 0 && with_copy <= (size_t)150 * 1024);
}
// wavefronts per workgroup of k_bin_sort: four, fewer where their tables would not fit in LDS
int bin_sort_waves(const DeviceState &st)
{
    int nw = 4;
    while (nw > 1 && bin_sort_lds_for(st, nw, bin_sort_stage(st)) > (size_t)150 * 1024) nw >>= 1;
    return nw;
}
size_t bin_sort_lds_bytes(const DeviceState &st)
{
    return bin_sort_lds_for(st, bin_sort_waves(st), bin_sort_stage(st));
}
static size_t sampler_lds_for(const DeviceState &st, int lanes)
{
    return sampler_lds_bytes(st.n_bin * st.n_class, st.n_class, st.max_appends,
                             lanes == 64 ? 64 : PMCX_WIN);
}
size_t sampler_lds_bytes_per_cell(const DeviceState &st)
{
    return sampler_lds_for(st, 64); // upper bound over the lane variants
}
// The sampler's form.  1: one lane per cell with the draw split off the sweep (k_coag_lane<.., PRE>;
// one class, Brownian); 3: one lane per cell, drawing as it sweeps; 2: that in its
// two-wavefront form; 8, 16, 64: lanes per cell (k_coag_sample).  pmcx_set_sampler_variant overrides
// (tests hold every form to the oracle at small sizes).
static bool pre_available(const DeviceState &st, int ktype)
{
    return st.n_class == 1 && ktype == KERNEL_BROWN && (st.n_bin & 3) == 0 && st.pre != nullptr &&
           seq_lds_bytes(st.n_bin, st.n_class, 32) <= 64 * 1024;
}
// several classes: the screen and the sweep of its list (k_coag_screen, k_coag_lane<.., LISTED>)
static bool listed_available(const DeviceState &st)
{
    return st.n_class > 1 && st.n_class <= 255 && st.n_bin <= 255 && (st.n_bin & 3) == 0 && st.slist != nullptr &&
           st.n_class * st.n_bin < 65535 && 4 * screen_lds_per_wave(st.n_class * st.n_bin, st.n_class) <= (size_t)150 * 1024;
}
int sampler_lanes(const DeviceState &st, int ktype)
{
    const int fl = st.force_lanes;
    const bool seq_fits = seq_lds_bytes(st.n_bin, st.n_class) + SEQ_SHARED_BYTES <= 64 * 1024;
    if ((fl == 0 || fl == 5) && listed_available(st)) return 5;
    if (fl == 1) return pre_available(st, ktype) ? 1 : (seq_fits ? 3 : 64);
    if (fl == 4) return pre_available(st, ktype) ? 4 : (seq_fits ? 3 : 64);   // (k_coag_grp: the configuration PRE runs on)
    if ((fl == 2 || fl == 3) && seq_fits) return fl;
    if (fl == 8 || fl == 16 || fl == 64) return fl;
    // One lane per cell with the draw split off wherever a wavefront's own chain is the floor --
    // measured (profiles/r3b_sampler_forms.txt, ms per launch against the best other form): 3 000
    // cells 0.65 / 0.73, 12 000 0.69 / 0.85, 39 000 0.82 / 0.98, 75 000 1.20 / 1.40, 120 000 1.69 /
    // 1.91, 150 000 2.14 / 2.17-2.31; at the full domain the kernel runs at the rate the chip serves
    // random lines, the list costs lines of its own and the one-phase form wins (300 000: 4.1-4.2
    // against 3.6-3.95 depending on the box).
    if (pre_available(st, ktype) && (long)st.n_cells >= 600L && (long)st.n_cells <= 200000L) return 1;
    // one lane per cell once the cells alone fill the chip (256 CUs x 4 SIMDs x 64 lanes); while
    // its wavefronts would still find a SIMD each even twice as many, the two-wavefront form
    // (measured, scripts/gpu_split.sh: 1.29 against 1.83 ms at 39 000 cells, 1.57 against 2.12 at
    // 63 000; at 75 000 a third wavefront lands on some SIMDs and it loses, 2.55 against 2.18)
    if (seq_fits && (long)st.n_cells > 256L * 4L * 64L) return 3;
    if (seq_fits && (long)st.n_cells >= 24000L) return 2;
    if (sampler_lds_for(st, 16) > 4 * 1024) return 64;
    // 8 lanes per cell (8 cells per wavefront) is the throughput optimum; with few
    // cells the kernel is one wavefront's latency, which is shortest with more lanes
    // per cell: widen while every wavefront still finds a slot (256 CUs x 12).
    const long slots = 256L * 12L;
    if ((long)st.n_cells <= slots) return 64;
    if ((long)st.n_cells <= slots * 4) return 16;
    return 8;
}
size_t compact_lds_bytes(const DeviceState &st)
{
    return (size_t)4 * ((size_t)(st.cap + 63) / 64 + 1) * (sizeof(uint64_t) + sizeof(uint32_t)) + 64;
}

size_t diagnostics_lds_bytes(const DeviceState &st)
{
    const size_t base = sizeof(double) * (size_t)(5 * st.n_dbin + 1 + 256) +
                        sizeof(uint32_t) * (size_t)(4 * st.n_class * st.n_dbin + st.n_class);
    return ((base + 15) & ~(size_t)15) + species_tile_bytes(st.n_spec, 4);
}

// dynamic LDS above 64 KiB needs the per-function opt-in
template <typename K>
static inline void allow_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

// the cells with work for the kernel that follows, listed in st.work_list (count first)
static void list_cells(const DeviceState &st, int what, int p0, int p1, int cell_first, int n_do,
                       hipStream_t s)
{
    (void)hipMemsetAsync(st.work_list, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(k_list_cells, dim3((n_do + 255) / 256), dim3(256), 0, s, st, what, p0, p1,
                       cell_first, n_do, st.work_list);
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
                              int64_t dst_row_stride, hipStream_t s, int elem_stride)
{
    hipLaunchKernelGGL(k_scatter_f64, grid2(cap, n_up), dim3(256), 0, s, src, dst, part_offset,
                       n_up, cap, dst_cell_stride, src_row_stride, n_rows, dst_row_stride,
                       elem_stride);
    return hipGetLastError();
}
hipError_t launch_scatter_i32(const int32_t *src, int32_t *dst, const int64_t *part_offset,
                              int n_up, int cap, int64_t dst_cell_stride, int64_t src_row_stride,
                              int n_rows, int64_t dst_row_stride, hipStream_t s, int elem_stride)
{
    hipLaunchKernelGGL(k_scatter_i32, grid2(cap, n_up), dim3(256), 0, s, src, dst, part_offset,
                       n_up, cap, dst_cell_stride, src_row_stride, n_rows, dst_row_stride,
                       elem_stride);
    return hipGetLastError();
}
hipError_t launch_gather_f64(double *dst, const double *src, const int64_t *part_offset, int n_dn,
                             int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s, int elem_stride)
{
    hipLaunchKernelGGL(k_gather_f64, grid2(src_cell_stride, n_dn), dim3(256), 0, s, dst, src,
                       part_offset, src_cell_stride, dst_row_stride, n_rows, src_row_stride,
                       elem_stride);
    return hipGetLastError();
}
hipError_t launch_gather_i32(int32_t *dst, const int32_t *src, const int64_t *part_offset,
                             int n_dn, int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s, int elem_stride)
{
    hipLaunchKernelGGL(k_gather_i32, grid2(src_cell_stride, n_dn), dim3(256), 0, s, dst, src,
                       part_offset, src_cell_stride, dst_row_stride, n_rows, src_row_stride,
                       elem_stride);
    return hipGetLastError();
}
hipError_t launch_set_n_part(int32_t *n_part, int32_t *n_hi, const int64_t *part_offset, int n_up,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_set_n_part, dim3((n_up + 255) / 256), dim3(256), 0, s, n_part, n_hi,
                       part_offset, n_up);
    return hipGetLastError();
}
hipError_t launch_compact(const DeviceState &st, int cell_first, int n_cells_do, int num, int den,
                          hipStream_t s)
{
    allow_lds(k_compact, compact_lds_bytes(st));
    list_cells(st, LIST_COMPACT, num, den, cell_first, n_cells_do, s);
    const int n_wg = (n_cells_do + 3) / 4;
    hipLaunchKernelGGL(k_compact, dim3(n_wg < 8192 ? n_wg : 8192), dim3(256), compact_lds_bytes(st), s,
                       st, (const int32_t *)(st.work_list + 1), (const int32_t *)st.work_list);
    return hipGetLastError();
}
hipError_t launch_refresh_caches(const DeviceState &st, int cell_first, int n_cells_do,
                                 hipStream_t s)
{
    hipLaunchKernelGGL(k_refresh_caches, grid2(st.cap, n_cells_do), dim3(256), 0, s, st,
                       cell_first, n_cells_do);
    return hipGetLastError();
}
// margin < 0: whole tables for every cell.  Otherwise scratch (2 n_cells + 2 ints) takes the lists
// of the cells whose bounds no longer hold, and only those are computed: narrow ranges by
// small workgroups with a small LDS footprint (many per CU), the others at full size.
hipError_t launch_kernel_bounds(const DeviceState &st, int ktype, int margin, int32_t *scratch,
                                hipStream_t s)
{
    const size_t lds = kernel_bounds_lds_bytes(st.n_bin);
    allow_lds(k_kernel_bounds, lds);
    const int n_wg = st.n_cells < 8192 ? st.n_cells : 8192;
    if (margin < 0 || !scratch) {
        hipLaunchKernelGGL(k_kernel_bounds, dim3(n_wg), dim3(512), lds, s, st, ktype, st.n_bin,
                           (const int32_t *)nullptr, (const int32_t *)nullptr);
        return hipGetLastError();
    }
    (void)hipMemsetAsync(scratch, 0, 2 * sizeof(int32_t), s);
    hipLaunchKernelGGL(k_bounds_check, dim3((st.n_cells + 255) / 256), dim3(256), 0, s, st, ktype, margin,
                       scratch);
    const int small = BOUNDS_SMALL < st.n_bin ? BOUNDS_SMALL : st.n_bin;
    hipLaunchKernelGGL(k_kernel_bounds, dim3(st.n_cells < 32768 ? st.n_cells : 32768), dim3(256),
                       kernel_bounds_lds_bytes(small), s, st, ktype, small,
                       (const int32_t *)(scratch + 2), (const int32_t *)scratch);
    if (small < st.n_bin)
        hipLaunchKernelGGL(k_kernel_bounds, dim3(n_wg), dim3(512), lds, s, st, ktype, st.n_bin,
                           (const int32_t *)(scratch + 2 + st.n_cells), (const int32_t *)(scratch + 1));
    return hipGetLastError();
}
hipError_t launch_bin_sort(const DeviceState &st, hipStream_t s)
{
    allow_lds(k_bin_sort, bin_sort_lds_bytes(st));
    list_cells(st, LIST_SORT, 0, 0, 0, st.n_cells, s);
    hipLaunchKernelGGL(k_bin_sort, dim3(st.n_cells < 16384 ? st.n_cells : 16384), dim3(64 * bin_sort_waves(st)),
                       bin_sort_lds_bytes(st), s, st, (const int32_t *)(st.work_list + 1),
                       (const int32_t *)st.work_list, bin_sort_stage(st) ? 1 : 0);
    return hipGetLastError();
}
bool coag_sample_splits_draw(const DeviceState &st, int ktype)
{
    const int L = sampler_lanes(st, ktype);
    return L == 1 || L == 4;      // (the forms that leave cells to a second launch: pmcx_get_sampler_info counts them)
}
// Several classes: rounds of (screen, sweep of the list) until no cell is left -- a cell is left when its
// list ran full or a group outgrew the length the list was made for; every round settles at least one
// slot with samples or SLIST_CAP - 1 listed ones, so the loop ends.  Two rounds are launched blind (the
// second finds its work list empty nearly always and returns at once), then the count is looked at.
static int g_listed_rounds = 0; // rounds the last call took (pmcx_get_sampler_info)
int coag_listed_rounds() { return g_listed_rounds; }
int coag_sample_form(const DeviceState &st, int ktype) { return sampler_lanes(st, ktype); }
static hipError_t launch_coag_listed(const DeviceState &st, int ktype, double del_t, uint64_t seed, uint32_t step,
                                     hipStream_t s)
{
    const int G = st.n_class * st.n_bin;
    const size_t lds_s = 4 * screen_lds_per_wave(G, st.n_class);
    allow_lds(k_coag_screen, lds_s);
    const size_t lds_l = seq_lds_bytes(st.n_bin, 0, 0);
    const int n_wg = (st.n_cells + 3) / 4;
    const dim3 grid_s(n_wg < 16384 ? n_wg : 16384), grid_l((st.n_cells + 63) / 64);
    int32_t *in = nullptr, *out = st.work2;
    for (int round = 0; round < 1000; round++) {
        (void)hipMemsetAsync(out, 0, sizeof(int32_t), s);
        const int32_t *il = in ? in + 1 : nullptr;
        hipLaunchKernelGGL(k_coag_screen, grid_s, dim3(256), lds_s, s, st, del_t, seed, step, il, (const int32_t *)in);
        if (ktype == KERNEL_BROWN)
            hipLaunchKernelGGL((k_coag_lane<false, false, KERNEL_BROWN, false, false, SEQ_W, true>), grid_l, dim3(64), lds_l, s,
                               st, ktype, del_t, seed, step, st.seq_ovf, il, in, out);
        else
            hipLaunchKernelGGL((k_coag_lane<false, false, 0, false, false, SEQ_W, true>), grid_l, dim3(64), lds_l, s,
                               st, ktype, del_t, seed, step, st.seq_ovf, il, in, out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (round & 1) {
            int32_t left = 0;
            e = hipMemcpyAsync(&left, out, sizeof(int32_t), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
            if (left == 0) { g_listed_rounds = round + 1; return hipSuccess; }
        }
        in = out;
        out = (out == st.work2) ? st.work3 : st.work2;
    }
    return hipErrorLaunchFailure; // (not reached: every round settles slots)
}

static hipError_t launch_coag_sample_only(const DeviceState &st, int ktype, double del_t, uint64_t seed,
                                          uint32_t step, hipStream_t s)
{
    const size_t shared = (sizeof(double) * (size_t)(st.n_bin + 1) + 15) & ~(size_t)15;
    const int L = sampler_lanes(st, ktype);
    if (L == 2) { // lane per cell, two wavefronts (evaluator, drawer) per 64 cells
        const dim3 grid((st.n_cells + 63) / 64), block(128);
        const size_t lds = seq_lds_bytes(st.n_bin, st.n_class) + SEQ_SHARED_BYTES;
        const bool simple = st.n_class == 1;
        if (simple && ktype == KERNEL_BROWN)
            hipLaunchKernelGGL((k_coag_lane<false, true, KERNEL_BROWN, true>), grid, block, lds, s, st,
                               ktype, del_t, seed, step, st.seq_ovf, (const int32_t *)nullptr, (int32_t *)nullptr);
        else if (ktype == KERNEL_BROWN)
            hipLaunchKernelGGL((k_coag_lane<false, false, KERNEL_BROWN, true>), grid, block, lds, s, st,
                               ktype, del_t, seed, step, st.seq_ovf, (const int32_t *)nullptr, (int32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_coag_lane<false, false, 0, true>), grid, block, lds, s, st, ktype,
                               del_t, seed, step, st.seq_ovf, (const int32_t *)nullptr, (int32_t *)nullptr);
        return hipGetLastError();
    }
    if (L == 5) return launch_coag_listed(st, ktype, del_t, seed, step, s);
    if (L == 4) { // the cells of a workgroup grouped by what they do next; what it hands over: the one-phase kernel
        (void)hipMemsetAsync(st.work2, 0, sizeof(int32_t), s);
        const size_t lds_g = grp_lds_bytes(st.n_bin);
        allow_lds(k_coag_grp, lds_g);
        hipLaunchKernelGGL(k_coag_grp, dim3((st.n_cells + GRP_NC - 1) / GRP_NC), dim3(GRP_NC), lds_g, s, st, del_t, seed,
                           step, st.work2);
        hipLaunchKernelGGL((k_coag_lane<false, true, KERNEL_BROWN>), dim3((st.n_cells + 63) / 64), dim3(64),
                           seq_lds_bytes(st.n_bin, st.n_class), s, st, ktype, del_t, seed, step, st.seq_ovf,
                           (const int32_t *)(st.work2 + 1), st.work2);
        return hipGetLastError();
    }
    if (L == 1 || L == 3) {
        const bool stamps = st.force_stamps == 1;
        const dim3 grid((st.n_cells + 63) / 64), block(64);
        size_t lds = seq_lds_bytes(st.n_bin, st.n_class);
        // (diagnostic: wavefronts_per_group = n caps the wavefronts a SIMD holds at n by padding LDS)
        if ((st.force_wpb == 1 || st.force_wpb == 2) && lds < (size_t)160 * 1024 / (4 * st.force_wpb) - 1024)
            lds = (size_t)160 * 1024 / (4 * st.force_wpb) - 1024;
        const bool simple = st.n_class == 1;
        const int32_t *no_list = nullptr;
        int32_t *no_count = nullptr;
#define PMCX_LAUNCH_LANE(STAMPS_, SIMPLE_, KT_)                                                     \
    hipLaunchKernelGGL((k_coag_lane<STAMPS_, SIMPLE_, KT_>), grid, block, lds, s, st, ktype, del_t, seed, step, \
                       st.seq_ovf, no_list, no_count)
        // The draw split off the sweep (PRE) for the reference's configuration: one class, one
        // species, Brownian; the cells it does not finish are listed and finished by the kernel
        // without PRE right behind it (its blocks beyond the list return at once).
        if (coag_sample_splits_draw(st, ktype)) {
            (void)hipMemsetAsync(st.work2, 0, sizeof(int32_t), s);
            const size_t lds32 = lds + (seq_lds_bytes(st.n_bin, st.n_class, 32) - seq_lds_bytes(st.n_bin, st.n_class));
            if (stamps)
                hipLaunchKernelGGL((k_coag_lane<true, true, KERNEL_BROWN, false, true>), grid, block, lds32, s, st,
                                   ktype, del_t, seed, step, st.seq_ovf, no_list, st.work2);
            else
                hipLaunchKernelGGL((k_coag_lane<false, true, KERNEL_BROWN, false, true>), grid, block, lds32, s, st,
                                   ktype, del_t, seed, step, st.seq_ovf, no_list, st.work2);
            hipLaunchKernelGGL((k_coag_lane<false, true, KERNEL_BROWN>), grid, block, lds, s, st, ktype, del_t,
                               seed, step, st.seq_ovf, (const int32_t *)(st.work2 + 1), st.work2);
            return hipGetLastError();
        }
        if (stamps && simple && ktype == KERNEL_BROWN) PMCX_LAUNCH_LANE(true, true, KERNEL_BROWN);
        else if (stamps) PMCX_LAUNCH_LANE(true, false, 0);
        else if (simple && ktype == KERNEL_BROWN) PMCX_LAUNCH_LANE(false, true, KERNEL_BROWN);
        else if (ktype == KERNEL_BROWN) PMCX_LAUNCH_LANE(false, false, KERNEL_BROWN);
        else PMCX_LAUNCH_LANE(false, false, 0);
#undef PMCX_LAUNCH_LANE
        return hipGetLastError();
    }
    const size_t per = sampler_lds_for(st, L);
    const int gpw = 64 / L;
    // wavefronts per workgroup: 2 measured best
    int wpb = st.force_wpb ? st.force_wpb : 2;
    if (wpb != 1 && wpb != 2 && wpb != 4) wpb = 2;
    while (wpb > 1 && shared + per * wpb * gpw > 64 * 1024) wpb >>= 1;
    if (shared + per * wpb * gpw > 160 * 1024) return hipErrorInvalidConfiguration;
    const int cpb = wpb * gpw;
    const size_t lds = shared + per * cpb;
    const dim3 grid((st.n_cells + cpb - 1) / cpb), block(64 * wpb);
#define PMCX_LAUNCH_SAMPLER(LL)                                                                  \
    do {                                                                                         \
        allow_lds(k_coag_sample<LL>, lds);                                                       \
        hipLaunchKernelGGL((k_coag_sample<LL>), grid, block, lds, s, st, ktype, del_t, seed,     \
                           step, (uint32_t)per);                                                 \
    } while (0)
    if (L == 8) PMCX_LAUNCH_SAMPLER(8);
    else if (L == 16) PMCX_LAUNCH_SAMPLER(16);
    else PMCX_LAUNCH_SAMPLER(64);
#undef PMCX_LAUNCH_SAMPLER
    return hipGetLastError();
}
hipError_t launch_coag_sample(const DeviceState &st, int ktype, double del_t, uint64_t seed,
                              uint32_t step, hipStream_t s)
{
    const hipError_t e = launch_coag_sample_only(st, ktype, del_t, seed, step, s);
    if (e != hipSuccess) return e;
    if (st.n_spec > 1) {
        const int n_wg = (st.n_cells + 3) / 4;
        hipLaunchKernelGGL(k_vol_merge, dim3(n_wg < 16384 ? n_wg : 16384), dim3(256), 0, s, st);
    }
    if (!st.n_comp_max) return hipGetLastError();
    hipLaunchKernelGGL(k_comp_merge, dim3((st.n_cells + 63) / 64), dim3(64), 0, s, st);
    return hipGetLastError();
}

hipError_t launch_comp_upload(const DeviceState &st, int cell_first, int n_up, const int64_t *part_offset,
                              const int32_t *len, const int32_t *source, const double *time, hipStream_t s)
{
    hipLaunchKernelGGL(k_comp_upload, grid2(st.cap, n_up), dim3(256), 0, s, st, cell_first, part_offset,
                       len, source, time);
    return hipGetLastError();
}
hipError_t launch_comp_download(const DeviceState &st, int cell_first, int n_dn, const int64_t *part_offset,
                                int32_t *len, int32_t *source, double *time, hipStream_t s)
{
    hipLaunchKernelGGL(k_comp_download, grid2(st.cap, n_dn), dim3(256), 0, s, st, cell_first, part_offset,
                       len, source, time);
    return hipGetLastError();
}
hipError_t launch_comp_default(const DeviceState &st, int cell_first, int n_up, hipStream_t s)
{
    hipLaunchKernelGGL(k_comp_default, grid2(st.cap, n_up), dim3(256), 0, s, st, cell_first);
    return hipGetLastError();
}

hipError_t launch_rebalance(const DeviceState &st, int allow_doubling, int allow_halving,
                            uint64_t seed, uint32_t step, hipStream_t s)
{
    if (st.n_class == 1) { // the test is a comparison of two per-cell scalars: list the cells first
        list_cells(st, LIST_REBALANCE, allow_doubling, allow_halving, 0, st.n_cells, s);
        hipLaunchKernelGGL(k_rebalance, dim3(st.n_cells < 32768 ? st.n_cells : 32768), dim3(64), 0, s,
                           st, allow_doubling, allow_halving, seed, step,
                           (const int32_t *)(st.work_list + 1), (const int32_t *)st.work_list);
    } else {
        hipLaunchKernelGGL(k_rebalance, dim3(st.n_cells), dim3(64), 0, s, st, allow_doubling,
                           allow_halving, seed, step, (const int32_t *)nullptr, (const int32_t *)nullptr);
    }
    // (the cells a halving left more than 1/compact_den dead are compacted by whoever looks next:
    // mc_coag's pass in front of the sampler, the transport's after its movers have left -- when
    // holes are closed changes no result, and three launches a step are three launches)
    return hipGetLastError();
}
hipError_t launch_diagnostics(const DeviceState &st, double *tot_num, double *tot_mass,
                              double *num_dist, double *mass_dist, hipStream_t s)
{
    allow_lds(k_diagnostics, diagnostics_lds_bytes(st));
    hipLaunchKernelGGL(k_diagnostics, dim3(st.n_cells), dim3(256), diagnostics_lds_bytes(st), s, st,
                       tot_num, tot_mass, num_dist, mass_dist);
    return hipGetLastError();
}
hipError_t launch_diagnostics_ext(const DeviceState &st, double *species_mass, double *source_num,
                                  double *pm, hipStream_t s)
{
    const size_t lds = ((sizeof(double) * (size_t)(st.n_spec + st.n_source + 3) + 15) & ~(size_t)15) +
                       species_tile_bytes(st.n_spec, 4);
    allow_lds(k_diagnostics_ext, lds);
    hipLaunchKernelGGL(k_diagnostics_ext, dim3(st.n_cells), dim3(256), lds, s, st, species_mass,
                       source_num, pm);
    return hipGetLastError();
}
hipError_t launch_generate(const DeviceState &st, const GenSpecDev &g, int n, uint64_t seed,
                           double create_time, hipStream_t s)
{
    hipLaunchKernelGGL(k_generate, grid2(n, st.n_cells), dim3(256), 0, s, st, g, n, seed,
                       create_time);
    return hipGetLastError();
}
hipError_t launch_scenario_update(const DeviceState &st, double dt, double create_time, uint64_t seed,
                                  uint32_t step, const GenSpecDev *em, const double *em_conc,
                                  const double *dilution_rate, const GenSpecDev *bg,
                                  const double *bg_conc, int32_t *n_emit, int32_t *n_dil_in,
                                  int32_t *n_dil_out, hipStream_t s, uint32_t emit_kind,
                                  const double *height, const double *old_height)
{
    AeroDistDev e{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, b = e;
    if (em) e = AeroDistDev{em->n_mode, em_conc, em->ln_gm_radius, em->ln_gsd, em->vol_frac,
                            em->mode_class, em->mode_source, em->per_cell};
    if (bg) b = AeroDistDev{bg->n_mode, bg_conc, bg->ln_gm_radius, bg->ln_gsd, bg->vol_frac,
                            bg->mode_class, bg->mode_source, bg->per_cell};
    if (st.prep_on) (void)hipMemsetAsync(st.prep_resume, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(k_scenario_update<false>, dim3(st.n_cells), dim3(64), 0, s, st, dt, create_time,
                       seed, step, e, dilution_rate, b, n_emit, n_dil_in, n_dil_out, emit_kind, height, old_height);
    if (st.prep_on) // the cells that stopped at a mode whose weight wants preparing: the rest of their update
        hipLaunchKernelGGL(k_scenario_update<true>, dim3(st.n_cells < 2048 ? st.n_cells : 2048), dim3(64), 0, s, st, dt,
                           create_time, seed, step, e, dilution_rate, b, n_emit, n_dil_in, n_dil_out, emit_kind, height,
                           old_height);
    return hipGetLastError();
}
// pmcx_set_auto_grow: how many slots the adds of a scenario update can need (an upper bound)
__global__ __launch_bounds__(256) void k_adds_bound(DeviceState st, double dt, double elapsed_time, AeroDistDev emission,
                                                    const double *dilution_rate, AeroDistDev background,
                                                    const double *height, const double *old_height, int32_t *out)
{
    int need = 0;
    // With the weight preparation on a mode's sample is drawn at the weight the preparation leaves: a class
    // scaled to (n + mean cf) / ideal of its weight takes mean ideal / (n + mean cf) <= ideal / cf particles --
    // ideal x dt at elapsed_time 0 into an empty class, the formula's own corner
    const double cf = fmax(1.0, fmin(3600.0, elapsed_time)) / dt;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < st.n_cells; c += gridDim.x * blockDim.x) {
        if (!in_loop(st, c)) continue;
        const double *mag = st.mag + (int64_t)c * st.n_class;
        double b = (double)st.n_hi[c];
        if (st.prep_on)
            for (int q = 0; q < st.n_class; q++) b += st.n_ideal[(int64_t)c * st.n_class + q];
        auto upper = [](double mean) { return mean > 0.0 ? mean + 6.0 * sqrt(mean) + 12.0 : 0.0; };
        auto prepared = [&](double mean, int cl) { // the larger of the sample at the present and at the prepared weight
            if (!st.prep_on) return mean;
            const double at_ideal = st.n_ideal[(int64_t)c * st.n_class + cl] / cf;
            return mean > at_ideal ? mean : at_ideal;
        };
        for (int m = 0; m < emission.n_mode; m++) {
            const double nc = emission.num_conc[(int64_t)c * emission.n_mode + m];
            const int cl = emission.mode_class ? emission.mode_class[emission.mo(c) + m] : 0;
            if (nc > 0.0 && cl >= 0 && cl < st.n_class && mag[cl] > 0.0) b += upper(prepared(nc * dt / mag[cl], cl));
        }
        const double rate = dilution_rate ? dilution_rate[c] : 0.0;
        const bool grew = height && height[c] > old_height[c] && old_height[c] > 0.0;
        if ((rate > 0.0 || grew) && background.n_mode > 0) {
            double p_keep = exp(-rate * dt);
            if (grew) p_keep = p_keep * old_height[c] / height[c];
            const double p_loss = 1.0 - p_keep + 1e-12;
            for (int m = 0; m < background.n_mode; m++) {
                const double nc = background.num_conc[(int64_t)c * background.n_mode + m];
                const int cl = background.mode_class ? background.mode_class[background.mo(c) + m] : 0;
                if (nc > 0.0 && cl >= 0 && cl < st.n_class && mag[cl] > 0.0) b += upper(prepared(nc * p_loss / mag[cl], cl));
            }
        }
        const int bi = b < 1.0e9 ? (int)b + 1 : 1000000000;
        need = bi > need ? bi : need;
    }
    for (int d = 32; d >= 1; d >>= 1) { const int a = __shfl_xor(need, d, 64); need = a > need ? a : need; }
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], need);
}
hipError_t launch_adds_bound(const DeviceState &st, double dt, double elapsed_time, const GenSpecDev *em,
                             const double *em_conc, const double *dilution_rate, const GenSpecDev *bg,
                             const double *bg_conc, const double *height, const double *old_height, int32_t *out,
                             hipStream_t s)
{
    AeroDistDev e{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, b = e;
    if (em) e = AeroDistDev{em->n_mode, em_conc, em->ln_gm_radius, em->ln_gsd, em->vol_frac,
                            em->mode_class, em->mode_source, em->per_cell};
    if (bg) b = AeroDistDev{bg->n_mode, bg_conc, bg->ln_gm_radius, bg->ln_gsd, bg->vol_frac,
                            bg->mode_class, bg->mode_source, bg->per_cell};
    (void)hipMemsetAsync(out, 0, sizeof(int32_t), s);
    int g = (st.n_cells + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_adds_bound, dim3(g), dim3(256), 0, s, st, dt, elapsed_time, e, dilution_rate, b, height,
                       old_height, out);
    return hipGetLastError();
}
hipError_t launch_gas_update(const GasScenarioDev &g, double *mix_rat, const double *temp,
                             const double *pressure, const double *height, const double *old_height,
                             int n_cells, double dt, double elapsed_time, hipStream_t s,
                             const uint8_t *loop_mask)
{
    const int64_t n = (int64_t)n_cells * g.n_gas;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gas_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, mix_rat,
                       temp, pressure, height, old_height, n_cells, dt, elapsed_time, loop_mask);
    return hipGetLastError();
}
size_t transport_lds_bytes(const DeviceState &st, int nz)
{
    const size_t nd = (size_t)(4 + nz), C = (size_t)st.n_class;
    // (+ law 1: three doubles per class, a lane's copies per destination while it deals them)
    return sizeof(double) * (3 * C * nd + nd + 3 * C) + sizeof(int) * (2 * C * nd + nd + C) +
           sizeof(uint16_t) * 64 * nd + C * nd + 16;
}
static TransGrid make_grid(int nx, int nz, int ny, int periodic_x, int periodic_y,
                           const double *prob_h, const double *prob_v, const double *cell_volume,
                           void *outbox, int32_t *n_out, int32_t *n_arr, int32_t *n_exp,
                           uint64_t *dest_mask, int c_first)
{
    // a periodic direction one cell wide is closed: the cell would be its own neighbour
    periodic_x = periodic_x && nx > 1;
    periodic_y = periodic_y && ny > 1;
    return TransGrid{nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v, cell_volume, (int2 *)outbox,
                     n_out, n_arr, c_first, n_exp, dest_mask};
}
// decide: every cell draws its movers (outboxes, expected arrivals, destination masks)
hipError_t launch_transport_begin(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                                  int periodic_y, const double *prob_h, const double *prob_v,
                                  const double *cell_volume, void *outbox, int32_t *n_out,
                                  int32_t *n_arr, int32_t *n_exp, uint64_t *dest_mask, uint64_t seed,
                                  uint32_t step, hipStream_t s)
{
    const TransGrid tg = make_grid(nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v, cell_volume,
                                   outbox, n_out, n_arr, n_exp, dest_mask, 0);
    const size_t lds = transport_lds_bytes(st, nz);
    (void)hipMemsetAsync(n_exp, 0, sizeof(int32_t) * (size_t)st.n_cells, s);
    if (st.trans_law == 1) {
        allow_lds(k_trans_decide<true>, lds);
        hipLaunchKernelGGL(k_trans_decide<true>, dim3(st.n_cells), dim3(64), lds, s, st, tg, seed, step);
    } else {
        allow_lds(k_trans_decide<false>, lds);
        hipLaunchKernelGGL(k_trans_decide<false>, dim3(st.n_cells), dim3(64), lds, s, st, tg, seed, step);
    }
    return hipGetLastError();
}
// room, gather, remove on the rows [row_first, row_first + n_rows); ghost rows emptied
hipError_t launch_transport_finish(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                                   int periodic_y, const double *cell_volume, void *outbox,
                                   int32_t *n_out, int32_t *n_arr, int32_t *n_exp,
                                   uint64_t *dest_mask, int row_first, int n_rows, hipStream_t s)
{
    const int c_first = row_first * nx * nz, n_do = n_rows * nx * nz;
    const TransGrid tg = make_grid(nx, nz, ny, periodic_x, periodic_y, nullptr, nullptr, cell_volume,
                                   outbox, n_out, n_arr, n_exp, dest_mask, c_first);
    allow_lds(k_trans_room, compact_lds_bytes(st));
    hipLaunchKernelGGL(k_trans_room, dim3((st.n_cells + 3) / 4), dim3(256), compact_lds_bytes(st), s, st, tg);
    hipLaunchKernelGGL(k_trans_gather, dim3(n_do), dim3(64), 0, s, st, tg);
    hipLaunchKernelGGL(k_trans_remove, dim3(n_do), dim3(64), 0, s, st, tg);
    // the cells the movers left more than 1/compact_den dead
    if (hipError_t ce = launch_compact(st, c_first, n_do, 1, st.compact_den, s); ce != hipSuccess) return ce;
    if (n_rows != ny) { // a slab: rows 0 and ny - 1 are ghosts
        const Strip a{0, 1, nx * nz}, b{(ny - 1) * nx * nz, 1, nx * nz};
        hipLaunchKernelGGL(k_halo_clear, dim3((a.n + b.n + 255) / 256), dim3(256), 0, s, st, tg, a, b);
    }
    if (st.trans_cut_x) { // cut in x as well: columns 0 and nx - 1 are ghosts
        const Strip a{0, nx, ny * nz}, b{nx - 1, nx, ny * nz};
        hipLaunchKernelGGL(k_halo_clear, dim3((a.n + b.n + 255) / 256), dim3(256), 0, s, st, tg, a, b);
    }
    return hipGetLastError();
}
hipError_t launch_transport(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                            int periodic_y, const double *prob_h, const double *prob_v,
                            const double *cell_volume, void *outbox, int32_t *n_out, int32_t *n_arr,
                            int32_t *n_exp, uint64_t *dest_mask, uint64_t seed, uint32_t step,
                            hipStream_t s)
{
    hipError_t e = launch_transport_begin(st, nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v,
                                          cell_volume, outbox, n_out, n_arr, n_exp, dest_mask, seed,
                                          step, s);
    if (e != hipSuccess) return e;
    return launch_transport_finish(st, nx, nz, ny, periodic_x, periodic_y, cell_volume, outbox, n_out,
                                   n_arr, n_exp, dest_mask, 0, ny, s);
}
hipError_t launch_boundary_copy(const DeviceState &st, int n_pairs, const int32_t *from, const int32_t *to,
                                hipStream_t s)
{
    if (n_pairs > 0) hipLaunchKernelGGL(k_boundary_copy, dim3(n_pairs), dim3(64), 0, s, st, n_pairs, from, to);
    return hipGetLastError();
}
hipError_t launch_boundary_resample(const DeviceState &st, const int32_t *action, const GenSpecDev &bg,
                                    const double *bg_conc, const double *density, double create_time,
                                    uint64_t seed, uint32_t step, int32_t *n_added, int32_t *n_removed,
                                    hipStream_t s)
{
    const AeroDistDev b{bg.n_mode, bg_conc, bg.ln_gm_radius, bg.ln_gsd, bg.vol_frac, bg.mode_class,
                        bg.mode_source, bg.per_cell};
    hipLaunchKernelGGL(k_boundary_resample, dim3(st.n_cells), dim3(64), 0, s, st, action, b, density,
                       create_time, seed, step, n_added, n_removed);
    return hipGetLastError();
}
hipError_t launch_preweight_slab_begin(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                       int j_global0, int global_ny, const double *prob_h,
                                       const double *prob_v, const double *cell_volume, double *old,
                                       double *edge_out, hipStream_t s)
{
    PreweightArgs a{};
    a.nx = nx; a.nz = nz; a.ny = ny; a.ideal = ideal; a.prob_h = prob_h; a.prob_v = prob_v;
    a.cell_volume = cell_volume; a.old = old; a.slab = 1; a.j_global0 = j_global0; a.global_ny = global_ny;
    a.edge_out = edge_out;
    launch_class_counts(st, s, true);
    hipLaunchKernelGGL(k_preweight_old, dim3((unsigned)(((int64_t)st.n_cells * st.n_class + 63) / 64)), dim3(64), 0, s, st, a);
    const int64_t n = 2 * (int64_t)nx * nz * st.n_class + (st.trans_cut_x ? 2 * (int64_t)ny * nz * st.n_class : 0);
    hipLaunchKernelGGL(k_preweight_edge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, a);
    return hipGetLastError();
}
hipError_t launch_preweight_slab_finish(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                        int j_global0, int global_ny, int lo_edge, int hi_edge,
                                        const double *prob_h, const double *prob_v,
                                        const double *cell_volume, double *old, double *expect,
                                        const double *gin_lo, const double *gin_hi, int32_t *n_removed,
                                        uint64_t seed, uint32_t step, hipStream_t s, int w_edge,
                                        int e_edge, const double *gin_w, const double *gin_e)
{
    PreweightArgs a{};
    a.nx = nx; a.nz = nz; a.ny = ny; a.ideal = ideal; a.prob_h = prob_h; a.prob_v = prob_v;
    a.cell_volume = cell_volume; a.old = old; a.expect = expect; a.n_removed = n_removed;
    a.slab = 1; a.lo_edge = lo_edge; a.hi_edge = hi_edge; a.j_global0 = j_global0; a.global_ny = global_ny;
    a.gin_lo = gin_lo; a.gin_hi = gin_hi;
    a.w_edge = w_edge; a.e_edge = e_edge; a.gin_w = gin_w; a.gin_e = gin_e;
    const int64_t n = (int64_t)st.n_cells * st.n_class;
    hipLaunchKernelGGL(k_preweight_new, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, a);
    hipLaunchKernelGGL(k_preweight_scale, dim3(st.n_cells), dim3(64), 0, s, st, a, seed, step);
    return hipGetLastError(); // (the cells thinned past 1/compact_den dead: the transport's compaction, after the movers have left)
}
hipError_t launch_transport_preweight(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                      const double *prob_h, const double *prob_v,
                                      const double *cell_volume, double *old, double *expect,
                                      int32_t *n_removed, uint64_t seed, uint32_t step, hipStream_t s)
{
    PreweightArgs a{};
    a.nx = nx; a.nz = nz; a.ny = ny; a.ideal = ideal; a.prob_h = prob_h; a.prob_v = prob_v;
    a.cell_volume = cell_volume; a.old = old; a.expect = expect; a.n_removed = n_removed;
    launch_class_counts(st, s, true);
    hipLaunchKernelGGL(k_preweight_old, dim3((unsigned)(((int64_t)st.n_cells * st.n_class + 63) / 64)), dim3(64), 0, s, st, a);
    const int64_t n = (int64_t)st.n_cells * st.n_class;
    hipLaunchKernelGGL(k_preweight_new, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, a);
    hipLaunchKernelGGL(k_preweight_scale, dim3(st.n_cells), dim3(64), 0, s, st, a, seed, step);
    return hipGetLastError(); // (the cells thinned past 1/compact_den dead: the transport's compaction, after the movers have left)
}
size_t trans_probs_lds_bytes(const DeviceState &st, int nz)
{
    return sizeof(double) * ((size_t)2 * nz * nz + 8 * (size_t)nz + nz + 1 + (size_t)nz * st.n_class * 6 +
                             (size_t)nz * 4 + 3 * (size_t)nz) + sizeof(int) * (size_t)nz * st.n_class + 16;
}
hipError_t launch_transport_probs(const DeviceState &st, const TransFieldsArgs &a, hipStream_t s)
{
    launch_class_counts(st, s);
    TransFieldsDev f{};
    f.nx = a.nx; f.nz = a.nz; f.ny = a.ny; f.do_vertmix = a.do_vertmix; f.all_columns = a.all_columns;
    f.i_offset = a.i_offset; f.j_offset = a.j_offset; f.global_nx = a.global_nx; f.global_ny = a.global_ny;
    f.dt = a.dt; f.dx = a.dx; f.dy = a.dy;
    f.alt = a.alt; f.xkhh = a.xkhh; f.u_flux = a.u_flux; f.v_flux = a.v_flux; f.w_flux = a.w_flux;
    f.z = a.z; f.z_at_w = a.z_at_w; f.exch_h = a.exch_h;
    f.prob_h = a.prob_h; f.prob_v = a.prob_v; f.n_substeps = a.n_substeps; f.bad = a.bad;
    f.n_class_part = a.n_class_part; f.clamp_flag = a.clamp_flag;
    const size_t lds = trans_probs_lds_bytes(st, a.nz);
    allow_lds(k_trans_probs, lds);
    hipLaunchKernelGGL(k_trans_probs, dim3(a.nx * a.ny), dim3(256), lds, s, st, f);
    return hipGetLastError();
}
size_t halo_record_bytes(const DeviceState &st) { return halo_rec_bytes(st.n_spec, st.n_source, st.n_comp_max); }
hipError_t launch_halo_count(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                             Strip src, Strip dst, int32_t *counts, int32_t *offsets, hipStream_t s)
{
    const TransGrid tg = make_grid(nx, nz, ny, 0, 0, nullptr, nullptr, nullptr, outbox, n_out, nullptr,
                                   nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_halo_count, dim3(src.n), dim3(64), 0, s, st, tg, src, dst, counts);
    hipLaunchKernelGGL(k_halo_scan, dim3(1), dim3(64), 0, s, counts, offsets, src.n);
    return hipGetLastError();
}
hipError_t launch_halo_pack(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                            Strip src, Strip dst, const int32_t *offsets, void *recs, hipStream_t s)
{
    const TransGrid tg = make_grid(nx, nz, ny, 0, 0, nullptr, nullptr, nullptr, outbox, n_out, nullptr,
                                   nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_halo_pack, dim3(src.n), dim3(64), 0, s, st, tg, src, dst, offsets,
                       (unsigned char *)recs, (int)halo_rec_bytes(st.n_spec, st.n_source, st.n_comp_max));
    return hipGetLastError();
}
hipError_t launch_halo_unpack(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                              int32_t *n_exp, uint64_t *dest_mask, Strip ghost, Strip real, uint64_t mask_bit,
                              const int32_t *counts, int32_t *offsets, const void *recs, int32_t *bad,
                              hipStream_t s)
{
    const TransGrid tg = make_grid(nx, nz, ny, 0, 0, nullptr, nullptr, nullptr, outbox, n_out, nullptr,
                                   n_exp, dest_mask, 0);
    // (the direction codes are unambiguous on grids of at least 3 x 3 columns and at most 60 levels:
    // otherwise the gathers look at every source)
    const bool ok = 4 + nz <= 64 && nx >= 3 && ny >= 3;
    hipLaunchKernelGGL(k_halo_scan, dim3(1), dim3(64), 0, s, counts, offsets, ghost.n);
    hipLaunchKernelGGL(k_halo_unpack, dim3(ghost.n), dim3(64), 0, s, st, tg, ghost, real, counts, offsets,
                       (const unsigned char *)recs, (int)halo_rec_bytes(st.n_spec, st.n_source, st.n_comp_max),
                       ok ? mask_bit : ~0ull, bad);
    return hipGetLastError();
}
hipError_t launch_dry_deposition(const DeviceState &st, double dt, uint64_t seed, uint32_t step,
                                 const int32_t *surface, const double *per_cell6,
                                 int32_t *n_removed, int32_t *unrealistic, hipStream_t s)
{
    const size_t nc = (size_t)st.n_cells;
    DepArgs a{surface, per_cell6, per_cell6 + nc, per_cell6 + 2 * nc, per_cell6 + 3 * nc,
              per_cell6 + 4 * nc, per_cell6 + 5 * nc, n_removed, unrealistic};
    hipLaunchKernelGGL(k_dry_deposition, dim3(st.n_cells), dim3(256), 0, s, st, a, dt, seed, step);
    return hipGetLastError();
}
hipError_t launch_condense(const DeviceState &st, const double *kappa, const double *rel_humid,
                           double del_t, int n_sub, hipStream_t s)
{
    hipLaunchKernelGGL(k_condense, grid2(st.cap, st.n_cells), dim3(256), 0, s, st, kappa,
                       rel_humid, del_t, n_sub);
    return hipGetLastError();
}
// pmcx_grow / pmcx_set_auto_grow: what the fullest cell holds, or is about to
__global__ __launch_bounds__(256) void k_need_max(DeviceState st, const int32_t *n_exp, int32_t *out)
{
    int need = 0, hi = 0, bits = 0;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < st.n_cells; c += gridDim.x * blockDim.x) {
        int a = st.n_part[c] + (n_exp ? n_exp[c] : 0);
        const int b = st.n_hi[c];
        if (!n_exp && st.n_ideal) {
            // (no arrivals counted: the slots in use plus the classes' ideal counts -- what a doubling, or
            // the weight preparation of an add, can bring a cell in one call)
            double s = 0.0;
            for (int q = 0; q < st.n_class; q++) s += st.n_ideal[(int64_t)c * st.n_class + q];
            a = b + (s < 65536.0 ? (int)s + 1 : 65536);
        }
        need = a > need ? a : need;
        hi = b > hi ? b : hi;
        bits |= st.status[c];
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const int a = __shfl_xor(need, d, 64), b = __shfl_xor(hi, d, 64);
        need = a > need ? a : need;
        hi = b > hi ? b : hi;
        bits |= __shfl_xor(bits, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], need); atomicMax(&out[1], hi); atomicOr(&out[2], bits); }
}
__global__ __launch_bounds__(64) void k_trans_undo(DeviceState st, const int2 *outbox, const int32_t *n_out)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const int64_t cb = (int64_t)c * st.cap;
    if (st.key_valid[c]) {
        const int n = n_out[c];
        for (int e = lane; e < n; e += 64) {
            const int2 en = outbox[cb + e];
            if (en.x < 0) continue;                       // (an emitter's entry: its particle never left)
            const int sl = en.x & 0xFFFF;
            const int cl = st.n_class > 1 ? st.wclass[cb + sl] : 0;
            st.gkey[cb + sl] = key_of(st, st.hot[cb + sl].vtot, cl);
        }
    }
    if (lane == 0) atomicAnd(&st.status[c], ~(int)ST_OUTBOX_OVERFLOW);
}
hipError_t launch_transport_undo(const DeviceState &st, void *outbox, const int32_t *n_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_trans_undo, dim3(st.n_cells), dim3(64), 0, s, st, (const int2 *)outbox, n_out);
    return hipGetLastError();
}
hipError_t launch_need_max(const DeviceState &st, const int32_t *n_exp, int32_t *out2, hipStream_t s)
{
    (void)hipMemsetAsync(out2, 0, 3 * sizeof(int32_t), s);
    int g = (st.n_cells + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_need_max, dim3(g), dim3(256), 0, s, st, n_exp, out2);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void k_grow_tail(DeviceState st, int old_cap)
{
    for (int c = blockIdx.x; c < st.n_cells; c += gridDim.x) {
        const int64_t cb = (int64_t)c * st.cap;
        for (int i = old_cap + threadIdx.x; i < st.cap; i += blockDim.x) {
            st.hot[cb + i].vtot = -1.0;
            st.gkey[cb + i] = (uint16_t)0xFFFFu;
        }
    }
}
hipError_t launch_grow_tail(const DeviceState &st, int old_cap, hipStream_t s)
{
    hipLaunchKernelGGL(k_grow_tail, dim3(st.n_cells < 32768 ? st.n_cells : 32768), dim3(256), 0, s, st, old_cap);
    return hipGetLastError();
}
hipError_t launch_totals(const DeviceState &st, int64_t *out4, hipStream_t s)
{
    int g = (st.n_cells + 255) / 256;
    if (g > 128) g = 128; // four same-address atomics per workgroup: few workgroups
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
