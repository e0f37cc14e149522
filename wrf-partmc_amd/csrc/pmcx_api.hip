// pmcx_api.hip -- C-ABI (include/pmcx.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary for partmc_timestep's per-cell loop body
// (interface/wrf_pmc_driver.F90:169-254).  Owns device memory for a block of
// cells, moves packed host arrays in and out, launches the kernels on one HIP
// stream.  No CPU fallback: without a usable gfx950 device pmcx_create fails.
#include "../../include/pmcx.h"
#include "pmcx_kernels.h"
#include "pmcx_device_math.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace pmcx;

namespace {

thread_local std::string g_create_error;

constexpr double LN_10 = 2.302585092994046; // ln 10, correctly rounded (0x1.26bb1bbb55516p+1)

struct ProfEntry {
    double ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

} // namespace

struct pmcx_ctx {
    pmcx_config cfg{};
    hipStream_t stream = nullptr;
    DeviceState st{};
    std::string err;
    int64_t bytes = 0;
    std::vector<void *> allocs;
    // model tables, host copies
    std::vector<double> v_edge, v_samp, d_edge, d_width, density;
    double *d_kappa = nullptr;
    bool kappa_set = false;
    double *d_density = nullptr, *d_v_edge = nullptr, *d_v_samp = nullptr, *d_d_edge = nullptr,
           *d_d_width = nullptr;
    double *d_temp = nullptr, *d_pressure = nullptr, *d_rh = nullptr, *d_n_ideal = nullptr;
    // staging
    void *stage = nullptr;
    size_t stage_bytes = 0;
    int64_t *d_offsets = nullptr;
    size_t offsets_cap = 0;
    int64_t *d_totals = nullptr;
    void *d_outbox = nullptr;   // transport outboxes, allocated on first use
    int32_t *d_n_out = nullptr, *d_n_arr = nullptr;
    uint64_t *d_dest_mask = nullptr;
    int32_t *d_n_exp = nullptr;
    // slab transport across ranks (pmcx_transport_begin .. _finish)
    int32_t *d_halo = nullptr;      // [2 sides][counts(row_cells) | offsets(row_cells + 1)] + import scratch
    int halo_nx = 0, halo_nz = 0, halo_ny = 0, halo_px = 0; // halo_nx != 0: a transport is open
    int halo_row_cells = 0;         // row size d_halo was allocated for (kept across steps)
    int64_t halo_total[4] = {0, 0, 0, 0}; // records to export per side (after _export_size)
    size_t pw_edge_len = 0;               // doubles in d_pw_edge
    bool pw_x_set = false;                // pmcx_transport_preweight_x_terms has been called for the open preweight
    const double *pw_gin_w = nullptr, *pw_gin_e = nullptr;
    int pw_w_edge = 0, pw_e_edge = 0;
    double *d_halo_cv = nullptr;    // [n_cells] cell volumes of the open transport (own buffer: the
                                    // staging area may be reallocated between _begin and _finish)
    int32_t *d_halo_bad = nullptr;  // [1] records of an imported buffer that failed validation
    // probabilities built on the device (pmcx_transport_probs), consumed by the next transport
    double *d_prob_h = nullptr, *d_prob_v = nullptr;
    int32_t *d_prob_i = nullptr;    // [n_cells] bad | [n_cells n_class] particles per class | [n_cells] sub-steps
    double *d_preweight = nullptr;  // [2][n_cells][n_class] num_conc_old, expected_num_conc
    double *d_pw_edge = nullptr;    // [4][nx nz][n_class] slab form: terms sent out (2 sides), terms received
    int pw_nx = 0, pw_nz = 0, pw_ny = 0, pw_ideal = 0, pw_jg0 = 0, pw_gny = 0; // pw_nx != 0: a slab preweight is open
    int probs_nx = 0, probs_nz = 0, probs_ny = 0; // geometry they were built for (probs_nx != 0: ready)
    int64_t *d_acc = nullptr;   // running sums over calls: n_samp, n_coag, n_part, status
    double *d_diag = nullptr; // [n_cells*(2+2*n_dbin)]
    bool aero_data_set = false, env_set = false, n_ideal_set = false;
    // device blocks for the temporaries of one call (per-cell inputs, counters), kept between
    // calls: a hipMalloc/hipFree pair per call costs more than the kernels of the smaller stages
    struct TmpBlock { void *p; size_t bytes; bool used; };
    std::vector<TmpBlock> tmp_blocks;
    // gas phase (pmcx_set_gas_state, pmcx_set_gas_scenario)
    int32_t n_gas = 0;
    double *d_gas = nullptr;            // [n_cells][n_gas] mix_rat, ppb
    GasScenarioDev gas_scn{};
    std::vector<void *> gas_scn_allocs; // the scenario's device arrays
    double *d_height = nullptr;         // [2][n_cells] height, old height of the current update
    uint8_t *d_loop_mask = nullptr;     // [n_cells] pmcx_set_loop_cells (allocated on first use)
    // pmcx_set_aero_scenario_modes: the mode parameters of the emission (0) and background (1)
    // distributions kept between calls (a dist passed with gm_radius == NULL refers to them)
    struct ResidentModes { GenSpecDev g{}; void *dd = nullptr, *di = nullptr; } res_modes[2];
    bool sampler_pre_ran = false; // the last pmcx_mc_coag ran the sampler with its draw split off
    int sampler_form = 0;         // ... in this form (pmcx_set_sampler_variant's numbering)
    bool cls_fresh = false;       // st.cls_cnt holds the populations as they are (set by pmcx_transport_probs, see HIP_TRY)
    int bounds_kernel = 0; // kernel type the kmax tables are valid for (0 = none)
    int bounds_margin = 8;     // bins above a cell's populated range its bounds cover (< 0: whole tables)
    bool bounds_dirty = true;  // something other than mc_coag changed a population since the last check
    int bounds_age = 0;        // mc_coag calls since the last check (coagulation moves the range up)
    // profiling
    bool prof = false;
    // pmcx_set_async: the per-step stage calls return without waiting for the device
    bool async_stages = false;
    // pmcx_set_auto_grow: the block's slots per cell grow like PartMC's aero_particle_array
    bool auto_grow = false;
    double grow_fill = 0.75, grow_factor = 1.5;
    int32_t *d_need = nullptr;  // [3] max n_part + n_exp, max n_hi, the status words or-ed
    int32_t n_grown = 0;        // times the block has grown
    // pmcx_timestep: the status word of the step is started once and its stages add their bits
    bool hold_status = false;
    int32_t *h_flags = nullptr; // page-locked: [0] pmcx_transport_probs' sub-step clamp, read at the next synchronisation
    std::map<std::string, ProfEntry> profile;
    std::vector<hipEvent_t> event_pool;
};
static inline void counts_stale(pmcx_ctx *ctx) { if (ctx) ctx->cls_fresh = false; }


namespace {

int fail(pmcx_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

// (any device call of any entry point voids the per-class counts pmcx_transport_probs left for the
// preweight that follows it at once: whatever came in between may have changed a population)
static inline void counts_stale(std::nullptr_t) {}
#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        counts_stale(ctx);                                                                  \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PMCX_ENOMEM : PMCX_EDEVICE,        \
                        "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(pmcx_ctx *ctx, T **p, size_t count, bool zero = true)
{
    const size_t bytes = sizeof(T) * (count ? count : 1);
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess)
        return fail(ctx, PMCX_ENOMEM, "hipMalloc of %zu bytes failed: %s", bytes,
                    hipGetErrorString(e));
    if (zero) {
        e = hipMemsetAsync(q, 0, bytes, ctx->stream);
        if (e != hipSuccess) return fail(ctx, PMCX_EDEVICE, "hipMemset: %s", hipGetErrorString(e));
    }
    ctx->allocs.push_back(q);
    ctx->bytes += (int64_t)bytes;
    *p = (T *)q;
    return PMCX_OK;
}

// a block of at least `bytes` for the duration of a call: released blocks are reused by later
// calls, whose uploads and kernels run behind this call's on the one stream (so a block is never
// overwritten before the kernel that read it has finished, also when the calls do not wait:
// pmcx_set_async)
hipError_t tmp_malloc(pmcx_ctx *ctx, void **p, size_t bytes)
{
    if (bytes == 0) bytes = 1;
    for (auto &b : ctx->tmp_blocks)
        if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096) { b.used = true; *p = b.p; return hipSuccess; }
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    ctx->tmp_blocks.push_back({q, bytes, true});
    ctx->bytes += (int64_t)bytes;
    *p = q;
    return hipSuccess;
}
void tmp_release(pmcx_ctx *ctx, void *p)
{
    for (auto &b : ctx->tmp_blocks)
        if (b.p == p) b.used = false;
}

// the end of a per-step stage call: wait for the device, unless the caller asked not to
static inline hipError_t stage_end(pmcx_ctx *ctx)
{
    return ctx->async_stages ? hipSuccess : hipStreamSynchronize(ctx->stream);
}

int ensure_stage(pmcx_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes) return PMCX_OK;
    if (ctx->stage) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->stage));
        ctx->bytes -= (int64_t)ctx->stage_bytes;
        ctx->stage = nullptr;
        ctx->stage_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    ctx->bytes += (int64_t)bytes;
    return PMCX_OK;
}

int ensure_offsets(pmcx_ctx *ctx, size_t n)
{
    if (n <= ctx->offsets_cap) return PMCX_OK;
    if (ctx->d_offsets) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->d_offsets));
    }
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_offsets, sizeof(int64_t) * n));
    ctx->offsets_cap = n;
    return PMCX_OK;
}

struct ProfScope {
    pmcx_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    ProfScope(pmcx_ctx *c, const char *n) : ctx(c), name(n)
    {
        if (!ctx->prof) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, ctx->stream);
    }
    ~ProfScope()
    {
        if (!ctx->prof || !a) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->profile[name].pending.emplace_back(a, b);
        ctx->profile[name].launches++;
    }
};

void prof_drain(pmcx_ctx *ctx)
{
    for (auto &kv : ctx->profile) {
        for (auto &pr : kv.second.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(pr.second);
            (void)hipEventElapsedTime(&ms, pr.first, pr.second);
            kv.second.ms += ms;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

} // namespace

extern "C" {

static int grow_if_needed(pmcx_ctx *ctx, const int32_t *n_exp);
static int grow_for_adds(pmcx_ctx *ctx, double dt, double elapsed_time, const GenSpecDev *em, const double *em_conc,
                         const double *dil, const GenSpecDev *bg, const double *bg_conc, const double *height,
                         const double *old_height);


void pmcx_config_default(pmcx_config *cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (int32_t)sizeof(pmcx_config);
    cfg->device = 0;
    cfg->n_cells = 1;
    cfg->cap = 2048;    /* num_particles default 1000 (registry.partmc:1), doubled */
    cfg->n_spec = 20;   /* aero_data.dat */
    cfg->n_class = 1;
    cfg->n_source = 0;
    cfg->n_bin = 80;    /* 10 bins per decade of radius */
    cfg->r_min = 1e-10;
    cfg->r_max = 1e-2;
    cfg->max_events = 1024;
    cfg->max_appends = 256; /* 4 B of LDS per entry and cell in the sampler */
    cfg->n_dbin = 100;  /* driver:783 */
    cfg->i_water = -1;
    cfg->d_min = 1e-9;
    cfg->d_max = 1e-3;
    cfg->cell_id_base = 0;
    cfg->compact_den = 2;
    cfg->const_k = 0.25 / (60.0 * 2.0e8); /* PartMC constant-kernel beta_0 [upstream] */
    cfg->additive_beta = 1000.0;          /* PartMC additive-kernel beta_1 [upstream] */
}

const char *pmcx_last_error(const pmcx_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int pmcx_create(const pmcx_config *cfg, pmcx_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, PMCX_EINVAL, "pmcx_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(pmcx_config))
        return fail(nullptr, PMCX_EINVAL, "pmcx_config size mismatch (%d vs %zu)",
                    cfg->struct_size, sizeof(pmcx_config));
    if (cfg->n_cells < 1 || cfg->cap < 2 || cfg->cap > 65280 || cfg->n_spec < 1 ||
        cfg->n_class < 1 || cfg->n_class > 255 || cfg->n_source < 0 || cfg->n_bin < 1 ||
        cfg->n_bin > 128 || !(cfg->r_min > 0) || !(cfg->r_max > cfg->r_min) ||
        cfg->max_events < 1 || cfg->max_appends < 1 || cfg->max_appends > 4096 ||
        cfg->n_dbin < 1 || !(cfg->d_min > 0) || !(cfg->d_max > cfg->d_min) ||
        cfg->n_class * cfg->n_bin > 65535 || cfg->i_water >= cfg->n_spec)
        return fail(nullptr, PMCX_EINVAL, "pmcx_create: configuration out of range");
    if (cfg->n_comp_max < 0 || cfg->n_comp_max > 8 ||
        (cfg->n_comp_max > 0 && (cfg->n_source < 1 || cfg->n_source > 255)))
        return fail(nullptr, PMCX_EINVAL,
                    "pmcx_create: n_comp_max = %d needs 0 <= n_comp_max <= 8 and 1 <= n_source <= 255",
                    cfg->n_comp_max);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || cfg->device >= n_dev)
        return fail(nullptr, PMCX_EDEVICE, "no HIP device %d (found %d): this library has no CPU path",
                    cfg->device, n_dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, PMCX_EDEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PMCX_EDEVICE, "device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    if (hipSetDevice(cfg->device) != hipSuccess)
        return fail(nullptr, PMCX_EDEVICE, "hipSetDevice(%d) failed", cfg->device);

    pmcx_ctx *ctx = new pmcx_ctx();
    ctx->cfg = *cfg;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PMCX_EDEVICE, "hipStreamCreate failed");
    }
    DeviceState &st = ctx->st;
    st.n_cells = cfg->n_cells; st.cap = cfg->cap; st.n_spec = cfg->n_spec;
    st.n_class = cfg->n_class; st.n_source = cfg->n_source; st.n_bin = cfg->n_bin;
    st.n_dbin = cfg->n_dbin; st.i_water = cfg->i_water;
    st.max_events = cfg->max_events; st.max_appends = cfg->max_appends;
    st.n_comp_max = cfg->n_comp_max;
    st.trans_law = 1; // (pmcx_set_transport_law)
    st.compact_den = cfg->compact_den ? (int32_t)cfg->compact_den : 2;
    st.pcap = sort_pcap(cfg->cap, cfg->n_class * cfg->n_bin);
    st.cell_id_base = cfg->cell_id_base;
    st.const_k = cfg->const_k; st.additive_beta = cfg->additive_beta;
    st.rho_min = 0; st.rho_max = 0;

    // sort grid (aero_sorted bin_grid) and diagnostics grid (driver:783)
    const int B = cfg->n_bin;
    ctx->v_edge.resize(B + 1); ctx->v_samp.resize(3 * B);
    const double l0 = std::log(cfg->r_min), l1 = std::log(cfg->r_max);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k <= B; k++) {
        const double r = std::exp(l0 + (l1 - l0) * (double)k / (double)B);
        ctx->v_edge[k] = (4.0 / 3.0) * pi * r * r * r;
    }
    for (int b = 0; b < B; b++) {
        ctx->v_samp[3 * b + 0] = ctx->v_edge[b];
        ctx->v_samp[3 * b + 1] = std::sqrt(ctx->v_edge[b] * ctx->v_edge[b + 1]);
        ctx->v_samp[3 * b + 2] = ctx->v_edge[b + 1];
    }
    st.log2_v0 = (float)std::log2(ctx->v_edge[0]);
    st.inv_dlog2 = (float)((double)B / (std::log2(ctx->v_edge[B]) - std::log2(ctx->v_edge[0])));
    // piecewise-linear log2 is off by at most 0.0861: keep the estimate within one bin
    st.bin_fast = (st.inv_dlog2 * 0.09f < 0.9f) ? 1 : 0;
    const int nd = cfg->n_dbin;
    ctx->d_edge.resize(nd + 1); ctx->d_width.resize(nd);
    const double m0 = std::log(cfg->d_min), m1 = std::log(cfg->d_max);
    for (int k = 0; k <= nd; k++)
        ctx->d_edge[k] = std::exp(m0 + (m1 - m0) * (double)k / (double)nd);
    for (int k = 0; k < nd; k++)
        ctx->d_width[k] = std::log(ctx->d_edge[k + 1]) - std::log(ctx->d_edge[k]);
    st.dlog2_d0 = (float)std::log2(cfg->d_min);
    st.dinv_dlog2 = (float)((double)nd / (std::log2(cfg->d_max) - std::log2(cfg->d_min)));
    st.dbin_fast = (st.dinv_dlog2 * 0.09f < 0.9f) ? 1 : 0;

    const size_t nc = (size_t)cfg->n_cells, cap = (size_t)cfg->cap;
    const size_t G = (size_t)cfg->n_class * B;
    int rc = PMCX_OK;
#define A_(ptr, count) if (rc == PMCX_OK) rc = dev_alloc(ctx, &(ptr), (count))
    A_(ctx->d_density, (size_t)cfg->n_spec);
    A_(ctx->d_kappa, (size_t)cfg->n_spec);
    A_(ctx->d_v_edge, (size_t)B + 1);
    A_(ctx->d_v_samp, (size_t)3 * B);
    A_(ctx->d_d_edge, (size_t)nd + 1);
    A_(ctx->d_d_width, (size_t)nd);
    A_(ctx->d_temp, nc); A_(ctx->d_pressure, nc); A_(ctx->d_rh, nc);
    A_(st.mag, nc * cfg->n_class);
    A_(ctx->d_n_ideal, nc * cfg->n_class);
    st.kmax_stride = (B * B + 3) & ~3;
    A_(st.kmax, nc * (size_t)st.kmax_stride);
    A_(st.kmax_rng, nc * 4); A_(st.kmax_env, nc * 2);
    A_(st.work_list, 2 * nc + 2);
    A_(st.n_part, nc); A_(st.n_hi, nc); A_(st.next_id, nc);
    A_(st.vol, nc * cfg->n_spec * cap);
    A_(st.hot, nc * cap);
    A_(st.wclass, nc * cap);
    A_(st.cold, nc * cap);
    A_(st.n_orig, nc * (size_t)(cfg->n_source > 1 ? cfg->n_source - 1 : 0) * cap + 1);
    A_(st.perm, nc * (size_t)st.pcap);
    A_(st.gcnt, nc * G); A_(st.goff, nc * (G + 1)); A_(st.sort_valid, nc);
    A_(st.gkey, nc * cap); A_(st.key_valid, nc); A_(st.evlog, nc * (size_t)EVLOG_CAP); A_(st.n_log, nc);
    A_(st.seq_ovf, nc * (2 * G + 2 * (size_t)cfg->max_appends));
    A_(st.n_events, nc); A_(st.n_samp, nc); A_(st.n_coag, nc); A_(st.status, nc);
    A_(st.resume, nc * 4); A_(st.work2, nc + 1);
    if (cfg->n_class == 1) { // the sampler's split-off draw (2 KB per cell)
        A_(st.pre, nc * 128);
    } else {                 // several classes: the list of possible slots (4 KB per cell), the lengths it was made for
        A_(st.slist, nc * (size_t)SLIST_CAP); A_(st.gnb, nc * G); A_(st.work3, nc + 1);
        A_(st.cls_cnt, nc * (size_t)cfg->n_class);
    }
    if (cfg->n_comp_max > 0) { // particle%component(:): 8 (1 + n_comp_max) bytes per slot
        A_(st.comp_src, nc * cap); A_(st.comp_time, nc * cap * (size_t)cfg->n_comp_max);
    }
    // the coagulations of a call in event order, a word each: what k_vol_merge (species volumes) and
    // k_comp_merge (component lists) apply behind the sampler
    if (cfg->n_comp_max > 0 || cfg->n_spec > 1) A_(st.mlog, nc * (size_t)cfg->max_events);
    A_(ctx->d_totals, (size_t)4);
    A_(ctx->d_acc, (size_t)4);
    A_(ctx->d_diag, nc * (size_t)(2 + 2 * nd));
#undef A_
    if (rc != PMCX_OK) {
        g_create_error = ctx->err;
        if (rc == PMCX_ENOMEM) {
            char more[256];
            snprintf(more, sizeof more, " (%d cells x %d slots, %.1f GB allocated before it; n_comp_max = %d "
                     "takes %d bytes of every slot)", cfg->n_cells, cfg->cap, (double)ctx->bytes / 1e9,
                     cfg->n_comp_max, cfg->n_comp_max > 0 ? 8 * (1 + cfg->n_comp_max) : 0);
            g_create_error += more;
        }
        pmcx_destroy(ctx);
        return rc;
    }
    st.density = ctx->d_density; st.v_edge = ctx->d_v_edge; st.v_samp = ctx->d_v_samp;
    st.d_edge = ctx->d_d_edge; st.d_width = ctx->d_d_width;
    st.temp = ctx->d_temp; st.pressure = ctx->d_pressure; st.n_ideal = ctx->d_n_ideal;

    hipError_t e = hipSuccess;
    auto up = [&](double *d, const std::vector<double> &h) {
        if (e == hipSuccess)
            e = hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice,
                               ctx->stream);
    };
    up(ctx->d_v_edge, ctx->v_edge); up(ctx->d_v_samp, ctx->v_samp);
    up(ctx->d_d_edge, ctx->d_edge); up(ctx->d_d_width, ctx->d_width);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        fail(nullptr, PMCX_EDEVICE, "table upload failed: %s", hipGetErrorString(e));
        pmcx_destroy(ctx);
        return PMCX_EDEVICE;
    }

    // shared-memory footprints must fit the CU's 160 KiB LDS
    const size_t lds_sort = bin_sort_lds_bytes(st), lds_samp = sampler_lds_bytes_per_cell(st);
    if (lds_sort > 160 * 1024 || lds_samp > 160 * 1024) {
        fail(nullptr, PMCX_EINVAL,
             "configuration needs %zu B (sort) / %zu B (sampler) of LDS per workgroup; limit 163840",
             lds_sort, lds_samp);
        pmcx_destroy(ctx);
        return PMCX_EINVAL;
    }
    *out = ctx;
    return PMCX_OK;
}

int pmcx_destroy(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_OK;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    prof_drain(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    for (void *p : ctx->allocs) (void)hipFree(p);
    for (void *p : ctx->gas_scn_allocs) (void)hipFree(p);
    for (auto &r : ctx->res_modes) { if (r.dd) (void)hipFree(r.dd); if (r.di) (void)hipFree(r.di); }
    for (auto &b : ctx->tmp_blocks) (void)hipFree(b.p);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->d_offsets) (void)hipFree(ctx->d_offsets);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PMCX_OK;
}

int pmcx_synchronize(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_flags && ctx->h_flags[0]) { // (what an asynchronous pmcx_transport_probs could not report)
        ctx->h_flags[0] = 0;
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: a column asked for more than 10^4 vertical "
                                      "sub-steps (z(2) - z(1) too small or exch_h too large; bit 16 of bad[]): "
                                      "the probabilities of that step were clamped");
    }
    return PMCX_OK;
}

int pmcx_set_async(pmcx_ctx *ctx, int32_t on)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (on && !ctx->h_flags) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_flags, sizeof(int32_t) * 4, hipHostMallocDefault));
        memset(ctx->h_flags, 0, sizeof(int32_t) * 4);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->async_stages = on != 0;
    return PMCX_OK;
}

// ---- growing particle arrays (PartMC's aero_particle_array grows; a block's slabs are re-pitched) ----

// One per-slot array of the block re-pitched from old_row to new_row elements per row: the new
// allocation first (so that a failed one leaves the block as it was), the copy and the swap later.
namespace {
struct Regrid {
    void **slot;        // where the context keeps the pointer
    size_t elem, rows, old_row, new_row;
    void *fresh;
};
}

int pmcx_grow(pmcx_ctx *ctx, int32_t new_cap)
{
    if (!ctx) return PMCX_EINVAL;
    pmcx_config &cf = ctx->cfg;
    DeviceState &st = ctx->st;
    new_cap = (new_cap + 63) & ~63;
    if (new_cap <= cf.cap) return PMCX_OK;
    if (new_cap > 65280)
        return fail(ctx, PMCX_EINVAL, "pmcx_grow: %d slots per cell asked for, a cell holds at most 65280 "
                                      "(16-bit list entries)", new_cap);
    if (ctx->halo_nx != 0 && ctx->d_halo)
        ; // (between pmcx_transport_begin and _finish: the outboxes are re-pitched with the rest)
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, oc = (size_t)cf.cap, nw = (size_t)new_cap;
    const size_t G = (size_t)cf.n_class * cf.n_bin;
    const size_t old_pcap = (size_t)st.pcap, new_pcap = (size_t)sort_pcap(new_cap, (int32_t)G);
    std::vector<Regrid> jobs;
    auto add = [&](void **slot, size_t elem, size_t rows, size_t old_row, size_t new_row) {
        if (*slot) jobs.push_back(Regrid{slot, elem, rows, old_row, new_row, nullptr});
    };
    add((void **)&st.vol, sizeof(double), nc, oc * cf.n_spec, nw * cf.n_spec);
    add((void **)&st.hot, sizeof(Hot), nc, oc, nw);
    add((void **)&st.wclass, sizeof(int32_t), nc, oc, nw);
    add((void **)&st.cold, sizeof(Cold), nc, oc, nw);
    if (cf.n_source > 1) add((void **)&st.n_orig, sizeof(int32_t), nc * (size_t)(cf.n_source - 1), oc, nw);
    add((void **)&st.perm, sizeof(uint16_t), nc, old_pcap, new_pcap);
    add((void **)&st.gkey, sizeof(uint16_t), nc, oc, nw);
    if (cf.n_comp_max > 0) {
        add((void **)&st.comp_src, sizeof(uint64_t), nc, oc, nw);
        add((void **)&st.comp_time, sizeof(double), nc, oc * (size_t)cf.n_comp_max, nw * (size_t)cf.n_comp_max);
    }
    add((void **)&ctx->d_outbox, sizeof(int64_t), nc, oc, nw);
    int64_t more = 0;
    for (Regrid &j : jobs) {
        const size_t bytes = j.elem * j.rows * j.new_row;
        if (hipMalloc(&j.fresh, bytes) != hipSuccess) {
            for (Regrid &k : jobs) if (k.fresh) (void)hipFree(k.fresh);
            return fail(ctx, PMCX_ENOMEM, "pmcx_grow: no memory for %d slots per cell (%d now): the block and its "
                                          "larger copy must both fit while it grows", new_cap, cf.cap);
        }
        more += (int64_t)bytes - (int64_t)(j.elem * j.rows * j.old_row);
    }
    for (Regrid &j : jobs) {
        // (zeroed like a fresh block's arrays: with one weight class nobody ever writes wclass)
        HIP_TRY(ctx, hipMemsetAsync(j.fresh, 0, j.elem * j.rows * j.new_row, ctx->stream));
        HIP_TRY(ctx, hipMemcpy2DAsync(j.fresh, j.elem * j.new_row, *j.slot, j.elem * j.old_row, j.elem * j.old_row,
                                      j.rows, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (Regrid &j : jobs) {
        for (void *&a : ctx->allocs) if (a == *j.slot) a = j.fresh;
        (void)hipFree(*j.slot);
        *j.slot = j.fresh;
    }
    ctx->bytes += more;
    const int old_cap = cf.cap;
    cf.cap = new_cap; st.cap = new_cap; st.pcap = (int32_t)new_pcap;
    HIP_TRY(ctx, launch_grow_tail(st, old_cap, ctx->stream));
    // (what was staged for the old pitch is rebuilt on use)
    ctx->n_grown++;
    return PMCX_OK;
}

int pmcx_set_auto_grow(pmcx_ctx *ctx, int32_t on, double fill, double factor)
{
    if (!ctx || (on && (!(fill > 0.0) || !(fill <= 1.0) || !(factor > 1.0) || !(factor <= 8.0))))
        return fail(ctx, PMCX_EINVAL, "pmcx_set_auto_grow: 0 < fill <= 1 < factor <= 8");
    if (on && !ctx->d_need) {
        int rc = dev_alloc(ctx, &ctx->d_need, (size_t)4);
        if (rc) return rc;
    }
    ctx->auto_grow = on != 0;
    if (on) { ctx->grow_fill = fill; ctx->grow_factor = factor; }
    return PMCX_OK;
}

int pmcx_get_growth(pmcx_ctx *ctx, int32_t *cap, int32_t *n_grown)
{
    if (!ctx) return PMCX_EINVAL;
    if (cap) *cap = ctx->cfg.cap;
    if (n_grown) *n_grown = ctx->n_grown;
    return PMCX_OK;
}

// pmcx_set_auto_grow: the fullest cell -- what it holds (n_exp null) or is about to hold after the
// arrivals its neighbours have counted for it (n_exp) -- read back (the call waits for the device
// here), and the block grown when that is beyond fill x cap: to factor x that, at least to what is
// needed, at most to 65 280.  What cannot fit even then is left to the kernels' status bit.
static int grow_if_needed(pmcx_ctx *ctx, const int32_t *n_exp)
{
    if (!ctx->auto_grow) return PMCX_OK;
    HIP_TRY(ctx, launch_need_max(ctx->st, n_exp, ctx->d_need, ctx->stream));
    int32_t h[3] = {0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_need, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // With the arrivals counted (transport): exact -- they find room among a cell's dead slots (k_trans_room
    // compacts), what counts is the live count plus the arrivals.  Without: a margin -- the fullest cell's
    // slots in use against fill x cap, and its slots in use plus its classes' ideal counts (what a
    // doubling or the weight preparation of an add can bring in one call; n_ideal set) against cap.
    const int cap = ctx->cfg.cap;
    int need = n_exp ? h[0] : h[1];
    bool must = need > cap;
    const bool should = (double)need > ctx->grow_fill * (double)cap;
    if (!n_exp && ctx->n_ideal_set && h[0] > cap) { must = true; need = h[0] > need ? h[0] : need; }
    if (!must && !should) return PMCX_OK;
    double want = ctx->grow_factor * (double)(n_exp && !must ? cap : need);
    if (!n_exp && must) want = 1.25 * (double)need > ctx->grow_factor * (double)h[1] ? 1.25 * (double)need : ctx->grow_factor * (double)h[1];
    if (want < (double)need) want = (double)need;
    if (want > 65280.0) want = 65280.0;
    int32_t new_cap = ((int32_t)want + 63) & ~63;
    if (new_cap > 65280) new_cap = 65280;
    if (new_cap <= cap) return PMCX_OK;
    const int rc = pmcx_grow(ctx, new_cap);
    if (rc == PMCX_ENOMEM && !must) { ctx->err.clear(); return PMCX_OK; } // (growing ahead of need is optional)
    return rc;
}

// pmcx_set_auto_grow: room for a scenario update's adds (emission, background, sea salt) -- an upper bound
// from the Poisson means of the samples (mean + 6 sigma + 12 a mode) and, with the weight preparation on,
// the classes' ideal counts; the block grows to 1.25 x the fullest cell's bound
static int grow_for_adds(pmcx_ctx *ctx, double dt, double elapsed_time, const GenSpecDev *em, const double *em_conc,
                         const double *dil, const GenSpecDev *bg, const double *bg_conc, const double *height,
                         const double *old_height)
{
    HIP_TRY(ctx, launch_adds_bound(ctx->st, dt, elapsed_time, em, em_conc, dil, bg, bg_conc, height, old_height,
                                   ctx->d_need, ctx->stream));
    int32_t need = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&need, ctx->d_need, sizeof need, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (need <= ctx->cfg.cap) return PMCX_OK;
    double want = 1.25 * (double)need;
    if (want > 65280.0) want = 65280.0;
    if ((int32_t)want <= ctx->cfg.cap) return PMCX_OK;
    return pmcx_grow(ctx, (int32_t)want);
}

// pmcx_set_auto_grow, law 1 of the transport: a source cell's outbox holds cap entries (one per mover, one
// per destination an emitter sends copies to).  Where k_trans_decide ran out of them the decision is
// taken back (the movers' sort keys restored), the block grown and the decision taken again: it is
// addressed by (seed, cell, step, rank), so the second time gives the first time's movers and the rest.
static int decide_with_room(pmcx_ctx *ctx, const std::function<hipError_t()> &begin)
{
    for (int round = 0;; round++) {
        HIP_TRY(ctx, begin());
        if (!ctx->auto_grow || ctx->st.trans_law != 1) return PMCX_OK;
        HIP_TRY(ctx, launch_need_max(ctx->st, nullptr, ctx->d_need, ctx->stream));
        int32_t h[3] = {0, 0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_need, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (!(h[2] & ST_OUTBOX_OVERFLOW) || ctx->cfg.cap >= 65280 || round >= 12) return PMCX_OK;
        double want = ctx->grow_factor * (double)ctx->cfg.cap;
        if (want > 65280.0) want = 65280.0;
        const int rc = pmcx_grow(ctx, (int32_t)want);
        if (rc) return rc;
        HIP_TRY(ctx, launch_transport_undo(ctx->st, ctx->d_outbox, ctx->d_n_out, ctx->stream));
    }
}

int pmcx_memory_bytes(pmcx_ctx *ctx, int64_t *bytes)
{
    if (!ctx || !bytes) return PMCX_EINVAL;
    *bytes = ctx->bytes;
    return PMCX_OK;
}

int pmcx_device_memory_info(int32_t device, int64_t *free_bytes, int64_t *total_bytes)
{
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess)
        return fail(nullptr, PMCX_EDEVICE, "pmcx_device_memory_info: no HIP device %d", device);
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return PMCX_OK;
}

int pmcx_set_aero_data(pmcx_ctx *ctx, const double *density, const double *kappa,
                       const double *molec_weight)
{
    (void)molec_weight;
    if (!ctx || !density) return fail(ctx, PMCX_EINVAL, "pmcx_set_aero_data: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ctx->density.assign(density, density + ctx->cfg.n_spec);
    double lo = density[0], hi = density[0];
    for (int s = 0; s < ctx->cfg.n_spec; s++) {
        if (!(density[s] > 0)) return fail(ctx, PMCX_EINVAL, "density[%d] must be positive", s);
        lo = density[s] < lo ? density[s] : lo;
        hi = density[s] > hi ? density[s] : hi;
    }
    ctx->st.rho_min = lo; ctx->st.rho_max = hi;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_density, density, sizeof(double) * ctx->cfg.n_spec,
                                hipMemcpyHostToDevice, ctx->stream));
    if (kappa) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kappa, kappa, sizeof(double) * ctx->cfg.n_spec,
                                    hipMemcpyHostToDevice, ctx->stream));
        ctx->kappa_set = true;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->aero_data_set = true;
    ctx->bounds_kernel = 0;
    // the bounds depend on the densities: no cell's table is current any more
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.kmax_rng, 0, sizeof(int32_t) * 4 * (size_t)ctx->cfg.n_cells,
                                ctx->stream));
    return PMCX_OK;
}

int pmcx_set_env(pmcx_ctx *ctx, const double *temp, const double *pressure,
                 const double *rel_humid)
{
    if (!ctx || !temp || !pressure) return fail(ctx, PMCX_EINVAL, "pmcx_set_env: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t b = sizeof(double) * (size_t)ctx->cfg.n_cells;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_temp, temp, b, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pressure, pressure, b, hipMemcpyHostToDevice, ctx->stream));
    if (rel_humid)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rh, rel_humid, b, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, stage_end(ctx));
    ctx->env_set = true;
    ctx->bounds_kernel = 0; // the bounds are those of the environment they were computed in
    return PMCX_OK;
}

// the lists of these cells no longer describe their particles (aero_state%valid_sort = .false.)
static hipError_t invalidate_sort(pmcx_ctx *ctx, int cell_first, int n, bool keys = true)
{
    ctx->bounds_dirty = true; // whatever changed the populations may have moved their bin ranges
    if (keys) { // the particles themselves changed: the sort keys are to be computed again
        hipError_t e = hipMemsetAsync(ctx->st.key_valid + cell_first, 0, sizeof(int32_t) * (size_t)n,
                                      ctx->stream);
        if (e != hipSuccess) return e;
    }
    return hipMemsetAsync(ctx->st.sort_valid + cell_first, 0, sizeof(int32_t) * (size_t)n, ctx->stream);
}

static int copy_cells_h2d(pmcx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}
static int copy_cells_d2h(pmcx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_set_weights(pmcx_ctx *ctx, const double *magnitude)
{
    if (!ctx || !magnitude) return fail(ctx, PMCX_EINVAL, "pmcx_set_weights: null argument");
    return copy_cells_h2d(ctx, ctx->st.mag, magnitude,
                          sizeof(double) * (size_t)ctx->cfg.n_cells * ctx->cfg.n_class);
}
int pmcx_get_weights(pmcx_ctx *ctx, double *magnitude)
{
    if (!ctx || !magnitude) return fail(ctx, PMCX_EINVAL, "pmcx_get_weights: null argument");
    return copy_cells_d2h(ctx, magnitude, ctx->st.mag,
                          sizeof(double) * (size_t)ctx->cfg.n_cells * ctx->cfg.n_class);
}
int pmcx_set_n_part_ideal(pmcx_ctx *ctx, const double *n_ideal)
{
    if (!ctx || !n_ideal) return fail(ctx, PMCX_EINVAL, "pmcx_set_n_part_ideal: null argument");
    ctx->n_ideal_set = true;
    return copy_cells_h2d(ctx, ctx->d_n_ideal, n_ideal,
                          sizeof(double) * (size_t)ctx->cfg.n_cells * ctx->cfg.n_class);
}
int pmcx_set_weight_preparation(pmcx_ctx *ctx, int32_t on, int32_t allow_doubling, int32_t allow_halving)
{
    if (!ctx) return fail(ctx, PMCX_EINVAL, "pmcx_set_weight_preparation: null context");
    if (on && !ctx->n_ideal_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_set_weight_preparation: pmcx_set_n_part_ideal must come first");
    if (on && ctx->cfg.n_class > 64)
        return fail(ctx, PMCX_EINVAL, "pmcx_set_weight_preparation: at most 64 weight classes (a class count per lane)");
    if (on && !ctx->st.prep_resume) {
        int rc = dev_alloc(ctx, &ctx->st.prep_resume, 1 + 5 * (size_t)ctx->cfg.n_cells);
        if (rc) return rc;
    }
    ctx->st.prep_on = on ? 1 : 0;
    ctx->st.prep_doubling = allow_doubling ? 1 : 0;
    ctx->st.prep_halving = allow_halving ? 1 : 0;
    return PMCX_OK;
}
int pmcx_set_transport_law(pmcx_ctx *ctx, int32_t law)
{
    if (!ctx || (law != 0 && law != 1)) return fail(ctx, PMCX_EINVAL, "pmcx_set_transport_law: law is 0 or 1");
    ctx->st.trans_law = law;
    return PMCX_OK;
}
int pmcx_set_next_id(pmcx_ctx *ctx, const int64_t *next_id)
{
    if (!ctx || !next_id) return fail(ctx, PMCX_EINVAL, "pmcx_set_next_id: null argument");
    return copy_cells_h2d(ctx, ctx->st.next_id, next_id, sizeof(int64_t) * (size_t)ctx->cfg.n_cells);
}
int pmcx_get_next_id(pmcx_ctx *ctx, int64_t *next_id)
{
    if (!ctx || !next_id) return fail(ctx, PMCX_EINVAL, "pmcx_get_next_id: null argument");
    return copy_cells_d2h(ctx, next_id, ctx->st.next_id, sizeof(int64_t) * (size_t)ctx->cfg.n_cells);
}
int pmcx_get_n_part(pmcx_ctx *ctx, int32_t *n_part)
{
    if (!ctx || !n_part) return fail(ctx, PMCX_EINVAL, "pmcx_get_n_part: null argument");
    return copy_cells_d2h(ctx, n_part, ctx->st.n_part, sizeof(int32_t) * (size_t)ctx->cfg.n_cells);
}

int pmcx_upload_cells(pmcx_ctx *ctx, int32_t cell_first, int32_t n_up, const int64_t *part_offset,
                      const double *vol, const int32_t *wclass, const int64_t *id,
                      const double *least_ct, const double *greatest_ct, const int32_t *n_orig)
{
    if (!ctx || !part_offset || !vol)
        return fail(ctx, PMCX_EINVAL, "pmcx_upload_cells: null argument");
    const pmcx_config &cf = ctx->cfg;
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    if (cell_first < 0 || n_up < 1 || cell_first + n_up > cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "cell range [%d,%d) outside [0,%d)", cell_first,
                    cell_first + n_up, cf.n_cells);
    if (part_offset[0] != 0) return fail(ctx, PMCX_EINVAL, "part_offset[0] must be 0");
    for (int k = 0; k < n_up; k++) {
        const int64_t n = part_offset[k + 1] - part_offset[k];
        if (n < 0 || n > cf.cap)
            return fail(ctx, PMCX_EINVAL, "cell %d holds %lld particles, capacity %d",
                        cell_first + k, (long long)n, cf.cap);
    }
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const int64_t nt = part_offset[n_up];
    const size_t cap = (size_t)cf.cap;
    DeviceState &st = ctx->st;
    int rc = ensure_offsets(ctx, (size_t)n_up + 1);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_offsets, part_offset, sizeof(int64_t) * ((size_t)n_up + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    const size_t per_part = sizeof(double) * (size_t)cf.n_spec + 4 + 8 + 8 + 8 +
                            sizeof(int32_t) * (size_t)cf.n_source;
    unsigned char *sp = nullptr;
    auto h2d = [&](const void *src, size_t bytes) -> void * {
        void *d = sp;
        (void)hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        sp += (bytes + 15) & ~(size_t)15;
        return d;
    };
    // staging offsets are 16-byte aligned; enlarge the buffer estimate accordingly
    rc = ensure_stage(ctx, per_part * (size_t)(nt > 0 ? nt : 1) + 16 * 8);
    if (rc) return rc;
    sp = (unsigned char *)ctx->stage;
    const int64_t c0 = cell_first;
    if (nt > 0) {
        const double *s_vol = (const double *)h2d(vol, sizeof(double) * cf.n_spec * (size_t)nt);
        // device layout [cap][n_spec]: species s of particle i at i * n_spec + s
        HIP_TRY(ctx, launch_scatter_f64(s_vol, st.vol + c0 * cf.n_spec * cap, ctx->d_offsets, n_up,
                                        cf.cap, (int64_t)cf.n_spec * cap, nt, cf.n_spec,
                                        (int64_t)1, ctx->stream, cf.n_spec));
        if (wclass) {
            const int32_t *s = (const int32_t *)h2d(wclass, sizeof(int32_t) * (size_t)nt);
            HIP_TRY(ctx, launch_scatter_i32(s, st.wclass + c0 * cap, ctx->d_offsets, n_up, cf.cap,
                                            (int64_t)cap, nt, 1, (int64_t)cap, ctx->stream));
        } else {
            HIP_TRY(ctx, hipMemsetAsync(st.wclass + c0 * cap, 0, sizeof(int32_t) * cap * n_up,
                                        ctx->stream));
        }
        if (id) {
            const double *s = (const double *)h2d(id, sizeof(int64_t) * (size_t)nt);
            HIP_TRY(ctx, launch_scatter_f64(s, (double *)&st.cold[c0 * cap].id, ctx->d_offsets, n_up,
                                            cf.cap, (int64_t)cap * 2, nt, 1, (int64_t)cap, ctx->stream, 2));
        }
        if (least_ct) {
            const double *s = (const double *)h2d(least_ct, sizeof(double) * (size_t)nt);
            HIP_TRY(ctx, launch_scatter_f64(s, &st.hot[c0 * cap].lct, ctx->d_offsets, n_up, cf.cap,
                                            (int64_t)cap * 4, nt, 1, (int64_t)cap, ctx->stream, 4));
        }
        if (greatest_ct) {
            const double *s = (const double *)h2d(greatest_ct, sizeof(double) * (size_t)nt);
            HIP_TRY(ctx, launch_scatter_f64(s, &st.hot[c0 * cap].gct, ctx->d_offsets, n_up, cf.cap,
                                            (int64_t)cap * 4, nt, 1, (int64_t)cap, ctx->stream, 4));
        }
        if (n_orig && cf.n_source > 0) {
            const int32_t *s =
                (const int32_t *)h2d(n_orig, sizeof(int32_t) * cf.n_source * (size_t)nt);
            // source 0 lives in the cold record (4 int32 per record), the rest in planes
            HIP_TRY(ctx, launch_scatter_i32(s, &st.cold[c0 * cap].n_orig0, ctx->d_offsets, n_up,
                                            cf.cap, (int64_t)cap * 4, nt, 1, (int64_t)cap, ctx->stream, 4));
            if (cf.n_source > 1)
                HIP_TRY(ctx, launch_scatter_i32(s + nt, st.n_orig + c0 * (cf.n_source - 1) * cap,
                                                ctx->d_offsets, n_up, cf.cap,
                                                (int64_t)(cf.n_source - 1) * cap, nt,
                                                cf.n_source - 1, (int64_t)cap, ctx->stream));
        }
    }
    HIP_TRY(ctx, launch_set_n_part(st.n_part + c0, st.n_hi + c0, ctx->d_offsets, n_up, ctx->stream));
    if (cf.n_comp_max > 0) HIP_TRY(ctx, launch_comp_default(st, cell_first, n_up, ctx->stream));
    HIP_TRY(ctx, launch_refresh_caches(st, cell_first, n_up, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, invalidate_sort(ctx, cell_first, n_up));
    return PMCX_OK;
}

int pmcx_download_cells(pmcx_ctx *ctx, int32_t cell_first, int32_t n_dn, const int64_t *part_offset,
                        double *vol, int32_t *wclass, int64_t *id, double *least_ct,
                        double *greatest_ct, int32_t *n_orig)
{
    if (!ctx || !part_offset) return fail(ctx, PMCX_EINVAL, "pmcx_download_cells: null argument");
    const pmcx_config &cf = ctx->cfg;
    if (cell_first < 0 || n_dn < 1 || cell_first + n_dn > cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "cell range outside the context");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    // the host view is dense: close any holes left by removals first
    HIP_TRY(ctx, launch_compact(ctx->st, cell_first, n_dn, 0, 1, ctx->stream));
    std::vector<int32_t> np((size_t)n_dn);
    HIP_TRY(ctx, hipMemcpyAsync(np.data(), ctx->st.n_part + cell_first, sizeof(int32_t) * n_dn,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n_dn; k++)
        if (part_offset[k + 1] - part_offset[k] != np[k])
            return fail(ctx, PMCX_EINVAL, "part_offset does not match n_part at cell %d",
                        cell_first + k);
    const int64_t nt = part_offset[n_dn];
    if (nt == 0) return PMCX_OK;
    const size_t cap = (size_t)cf.cap;
    DeviceState &st = ctx->st;
    int rc = ensure_offsets(ctx, (size_t)n_dn + 1);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_offsets, part_offset, sizeof(int64_t) * ((size_t)n_dn + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    const size_t biggest = sizeof(double) * (size_t)(cf.n_spec > 1 ? cf.n_spec : 1) * (size_t)nt;
    const size_t big2 = sizeof(int32_t) * (size_t)cf.n_source * (size_t)nt;
    rc = ensure_stage(ctx, biggest > big2 ? biggest : big2);
    if (rc) return rc;
    const int64_t c0 = cell_first;
    if (vol) {
        // with one species the volume of that species is the cached total (the lane-per-cell
        // sampler does not keep the plane current): gathered from the (vtot, mass) records
        if (cf.n_spec == 1)
            HIP_TRY(ctx, launch_gather_f64((double *)ctx->stage, &st.hot[c0 * cap].vtot, ctx->d_offsets,
                                           n_dn, (int64_t)cap * 4, nt, 1, (int64_t)cap, ctx->stream, 4));
        else
        HIP_TRY(ctx, launch_gather_f64((double *)ctx->stage, st.vol + c0 * cf.n_spec * cap,
                                       ctx->d_offsets, n_dn, (int64_t)cf.n_spec * cap, nt,
                                       cf.n_spec, (int64_t)1, ctx->stream, cf.n_spec));
        HIP_TRY(ctx, hipMemcpyAsync(vol, ctx->stage, sizeof(double) * cf.n_spec * (size_t)nt,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    // dev: first element of cell 0; es: element stride (2 for half of an interleaved pair)
    auto one_f64 = [&](double *host, const double *dev, int es = 1) -> int {
        HIP_TRY(ctx, launch_gather_f64((double *)ctx->stage, dev + c0 * cap * es, ctx->d_offsets, n_dn,
                                       (int64_t)cap * es, nt, 1, (int64_t)cap, ctx->stream, es));
        HIP_TRY(ctx, hipMemcpyAsync(host, ctx->stage, sizeof(double) * (size_t)nt,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return PMCX_OK;
    };
    if (id && (rc = one_f64((double *)id, (const double *)&st.cold[0].id, 2))) return rc;
    if (least_ct && (rc = one_f64(least_ct, &st.hot[0].lct, 4))) return rc;
    if (greatest_ct && (rc = one_f64(greatest_ct, &st.hot[0].gct, 4))) return rc;
    if (wclass) {
        HIP_TRY(ctx, launch_gather_i32((int32_t *)ctx->stage, st.wclass + c0 * cap, ctx->d_offsets,
                                       n_dn, (int64_t)cap, nt, 1, (int64_t)cap, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(wclass, ctx->stage, sizeof(int32_t) * (size_t)nt,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_orig && cf.n_source > 0) {
        HIP_TRY(ctx, launch_gather_i32((int32_t *)ctx->stage, &st.cold[c0 * cap].n_orig0,
                                       ctx->d_offsets, n_dn, (int64_t)cap * 4, nt, 1, (int64_t)cap,
                                       ctx->stream, 4));
        if (cf.n_source > 1)
            HIP_TRY(ctx, launch_gather_i32((int32_t *)ctx->stage + nt,
                                           st.n_orig + c0 * (cf.n_source - 1) * cap, ctx->d_offsets,
                                           n_dn, (int64_t)(cf.n_source - 1) * cap, nt,
                                           cf.n_source - 1, (int64_t)cap, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(n_orig, ctx->stage, sizeof(int32_t) * cf.n_source * (size_t)nt,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PMCX_OK;
}

int pmcx_upload_components(pmcx_ctx *ctx, int32_t cell_first, int32_t n_up, const int64_t *part_offset,
                           const int32_t *comp_len, const int32_t *comp_source, const double *comp_time)
{
    if (!ctx || !part_offset || !comp_len || !comp_source || !comp_time)
        return fail(ctx, PMCX_EINVAL, "pmcx_upload_components: null argument");
    const pmcx_config &cf = ctx->cfg;
    if (cf.n_comp_max < 1) return fail(ctx, PMCX_ESTATE, "pmcx_upload_components: the context was created with n_comp_max = 0");
    if (cell_first < 0 || n_up < 1 || cell_first + n_up > cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "cell range outside the context");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    std::vector<int32_t> np((size_t)n_up);
    HIP_TRY(ctx, hipMemcpyAsync(np.data(), ctx->st.n_hi + cell_first, sizeof(int32_t) * n_up,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n_up; k++) // (slots as pmcx_upload_cells laid them out: nothing removed since)
        if (part_offset[k + 1] - part_offset[k] != np[k])
            return fail(ctx, PMCX_EINVAL, "pmcx_upload_components: part_offset does not match the upload at cell %d",
                        cell_first + k);
    const int64_t nt = part_offset[n_up];
    if (nt == 0) return PMCX_OK;
    for (int64_t i = 0; i < nt; i++)
        for (int q = 0; q < comp_len[i] && q < cf.n_comp_max; q++)
            if (comp_source[i * cf.n_comp_max + q] < 0 || comp_source[i * cf.n_comp_max + q] >= 255)
                return fail(ctx, PMCX_EINVAL, "pmcx_upload_components: source %d of particle %lld outside [0,255) (a byte each)",
                            comp_source[i * cf.n_comp_max + q], (long long)i);
    int rc = ensure_offsets(ctx, (size_t)n_up + 1);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_offsets, part_offset, sizeof(int64_t) * ((size_t)n_up + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    const size_t K = (size_t)cf.n_comp_max, b_len = (sizeof(int32_t) * (size_t)nt + 15) & ~(size_t)15;
    const size_t b_src = (sizeof(int32_t) * K * (size_t)nt + 15) & ~(size_t)15, b_tim = sizeof(double) * K * (size_t)nt;
    rc = ensure_stage(ctx, b_len + b_src + b_tim);
    if (rc) return rc;
    unsigned char *sp = (unsigned char *)ctx->stage;
    HIP_TRY(ctx, hipMemcpyAsync(sp, comp_len, sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(sp + b_len, comp_source, sizeof(int32_t) * K * (size_t)nt, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(sp + b_len + b_src, comp_time, b_tim, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_comp_upload(ctx->st, cell_first, n_up, ctx->d_offsets, (const int32_t *)sp,
                                    (const int32_t *)(sp + b_len), (const double *)(sp + b_len + b_src), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_download_components(pmcx_ctx *ctx, int32_t cell_first, int32_t n_dn, const int64_t *part_offset,
                             int32_t *comp_len, int32_t *comp_source, double *comp_time)
{
    if (!ctx || !part_offset || !comp_len || !comp_source || !comp_time)
        return fail(ctx, PMCX_EINVAL, "pmcx_download_components: null argument");
    const pmcx_config &cf = ctx->cfg;
    if (cf.n_comp_max < 1) return fail(ctx, PMCX_ESTATE, "pmcx_download_components: the context was created with n_comp_max = 0");
    if (cell_first < 0 || n_dn < 1 || cell_first + n_dn > cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "cell range outside the context");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    HIP_TRY(ctx, launch_compact(ctx->st, cell_first, n_dn, 0, 1, ctx->stream)); // the host view is dense
    std::vector<int32_t> np((size_t)n_dn);
    HIP_TRY(ctx, hipMemcpyAsync(np.data(), ctx->st.n_part + cell_first, sizeof(int32_t) * n_dn,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n_dn; k++)
        if (part_offset[k + 1] - part_offset[k] != np[k])
            return fail(ctx, PMCX_EINVAL, "part_offset does not match n_part at cell %d", cell_first + k);
    const int64_t nt = part_offset[n_dn];
    if (nt == 0) return PMCX_OK;
    int rc = ensure_offsets(ctx, (size_t)n_dn + 1);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_offsets, part_offset, sizeof(int64_t) * ((size_t)n_dn + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    const size_t K = (size_t)cf.n_comp_max, b_len = (sizeof(int32_t) * (size_t)nt + 15) & ~(size_t)15;
    const size_t b_src = (sizeof(int32_t) * K * (size_t)nt + 15) & ~(size_t)15, b_tim = sizeof(double) * K * (size_t)nt;
    rc = ensure_stage(ctx, b_len + b_src + b_tim);
    if (rc) return rc;
    unsigned char *sp = (unsigned char *)ctx->stage;
    HIP_TRY(ctx, launch_comp_download(ctx->st, cell_first, n_dn, ctx->d_offsets, (int32_t *)sp,
                                      (int32_t *)(sp + b_len), (double *)(sp + b_len + b_src), ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(comp_len, sp, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(comp_source, sp + b_len, sizeof(int32_t) * K * (size_t)nt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(comp_time, sp + b_len + b_src, b_tim, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_update_kernel_bounds(pmcx_ctx *ctx, int32_t kernel_type)
{
    if (!ctx) return PMCX_EINVAL;
    if (!ctx->aero_data_set || !ctx->env_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data and pmcx_set_env must come first");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    {
        ProfScope ps(ctx, "kernel_bounds");
        HIP_TRY(ctx, launch_kernel_bounds(ctx->st, kernel_type, -1, nullptr, ctx->stream));
    }
    ctx->bounds_kernel = kernel_type;
    ctx->bounds_dirty = false;
    ctx->bounds_age = 0;
    return PMCX_OK;
}

int pmcx_set_kernel_bounds_margin(pmcx_ctx *ctx, int32_t margin)
{
    if (!ctx) return PMCX_EINVAL;
    if (margin > 127) return fail(ctx, PMCX_EINVAL, "pmcx_set_kernel_bounds_margin: at most 127 bins");
    ctx->bounds_margin = margin < 0 ? -1 : (margin < 2 ? 2 : margin);
    ctx->bounds_kernel = 0;
    return PMCX_OK;
}

static int check_kernel_type(pmcx_ctx *ctx, int32_t k)
{
    switch (k) {
    case PMCX_COAG_KERNEL_TYPE_SEDI: case PMCX_COAG_KERNEL_TYPE_ADDITIVE:
    case PMCX_COAG_KERNEL_TYPE_CONSTANT: case PMCX_COAG_KERNEL_TYPE_BROWN:
    case PMCX_COAG_KERNEL_TYPE_ZERO: case PMCX_COAG_KERNEL_TYPE_BROWN_SEDI:
        return PMCX_OK;
    default:
        return fail(ctx, PMCX_EINVAL, "unknown coag_kernel_type %d", k);
    }
}

int pmcx_sort_only(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ProfScope ps(ctx, "bin_sort");
    HIP_TRY(ctx, launch_bin_sort(ctx->st, ctx->stream));
    return PMCX_OK;
}

int pmcx_invalidate_sort(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells));
    return PMCX_OK;
}

int pmcx_set_transport_cuts(pmcx_ctx *ctx, int32_t cut_x, int32_t nx_local, int32_t i_global0, int32_t global_nx)
{
    if (!ctx) return PMCX_EINVAL;
    if (!cut_x) {
        ctx->st.trans_cut_x = 0; ctx->st.trans_i0 = 0; ctx->st.trans_gnx = 0;
        ctx->st.id_nx_local = 0; ctx->st.id_nx_global = 0;
        return PMCX_OK;
    }
    if (nx_local < 3 || global_nx < nx_local - 2 || ctx->cfg.n_cells % nx_local != 0)
        return fail(ctx, PMCX_EINVAL, "pmcx_set_transport_cuts: a block cut in x has nx_local >= 3 columns (two of "
                                      "them ghosts), n_cells a multiple of it, and global_nx >= nx_local - 2");
    ctx->st.trans_cut_x = 1;
    ctx->st.trans_i0 = i_global0;
    ctx->st.trans_gnx = global_nx;
    ctx->st.id_nx_local = nx_local;   // cell ids: rows of nx_local cells inside rows of global_nx
    ctx->st.id_nx_global = global_nx;
    return PMCX_OK;
}

int pmcx_set_transport_lateral(pmcx_ctx *ctx, int32_t on, int32_t j_global0, int32_t global_ny)
{
    if (!ctx) return PMCX_EINVAL;
    if (global_ny < 0 || (global_ny == 0 && j_global0 != 0))
        return fail(ctx, PMCX_EINVAL, "pmcx_set_transport_lateral: global_ny = 0 means the block is the domain (j_global0 = 0)");
    ctx->st.trans_lateral = on ? 1 : 0;
    ctx->st.trans_j0 = j_global0;
    ctx->st.trans_gny = global_ny;
    return PMCX_OK;
}

int pmcx_set_sampler_variant(pmcx_ctx *ctx, int32_t lanes, int32_t wavefronts_per_group, int32_t stamps)
{
    if (!ctx) return PMCX_EINVAL;
    if (!(lanes == 0 || lanes == 1 || lanes == 2 || lanes == 3 || lanes == 4 || lanes == 5 || lanes == 8 || lanes == 16 || lanes == 64) ||
        !(wavefronts_per_group == 0 || wavefronts_per_group == 1 || wavefronts_per_group == 2 ||
          wavefronts_per_group == 4) || (stamps != 0 && stamps != 1))
        return fail(ctx, PMCX_EINVAL, "pmcx_set_sampler_variant: lanes in {0,1,2,3,4,5,8,16,64}, "
                                      "wavefronts_per_group in {0,1,2,4}, stamps in {0,1}");
    ctx->st.force_lanes = lanes;
    ctx->st.force_wpb = wavefronts_per_group;
    ctx->st.force_stamps = stamps;
    return PMCX_OK;
}

int pmcx_get_sampler_info(pmcx_ctx *ctx, int32_t out[4])
{
    if (!ctx || !out) return fail(ctx, PMCX_EINVAL, "pmcx_get_sampler_info: null argument");
    out[0] = -1; out[1] = ctx->sampler_form; out[2] = ctx->sampler_form == 5 ? coag_listed_rounds() : 0; out[3] = 0;
    if (!ctx->st.work2 || !ctx->sampler_pre_ran) return PMCX_OK;
    return copy_cells_d2h(ctx, out, ctx->st.work2, sizeof(int32_t));
}

int pmcx_invalidate_lists(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false));
    return PMCX_OK;
}

int pmcx_get_keys(pmcx_ctx *ctx, int32_t cell, uint16_t *gkey, int32_t *key_valid, uint32_t *evlog,
                  int32_t *n_log)
{
    if (!ctx || cell < 0 || cell >= ctx->cfg.n_cells) return fail(ctx, PMCX_EINVAL, "pmcx_get_keys: bad argument");
    int rc = PMCX_OK;
    if (gkey) rc = copy_cells_d2h(ctx, gkey, ctx->st.gkey + (size_t)cell * ctx->cfg.cap, sizeof(uint16_t) * (size_t)ctx->cfg.cap);
    if (!rc && key_valid) rc = copy_cells_d2h(ctx, key_valid, ctx->st.key_valid + cell, sizeof(int32_t));
    if (!rc && evlog) rc = copy_cells_d2h(ctx, evlog, ctx->st.evlog + (size_t)cell * EVLOG_CAP, sizeof(uint32_t) * EVLOG_CAP);
    if (!rc && n_log) rc = copy_cells_d2h(ctx, n_log, ctx->st.n_log + cell, sizeof(int32_t));
    return rc;
}

int pmcx_mc_coag(pmcx_ctx *ctx, int32_t kernel_type, double del_t, uint64_t seed, uint32_t step)
{
    if (!ctx) return PMCX_EINVAL;
    int rc = check_kernel_type(ctx, kernel_type);
    if (rc) return rc;
    if (!(del_t >= 0)) return fail(ctx, PMCX_EINVAL, "del_t must be non-negative");
    if (ctx->bounds_margin < 0 && ctx->bounds_kernel != kernel_type) {
        rc = pmcx_update_kernel_bounds(ctx, kernel_type);
        if (rc) return rc;
    }
    if (!ctx->aero_data_set || !ctx->env_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data and pmcx_set_env must come first");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (!ctx->hold_status)
        HIP_TRY(ctx, hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * (size_t)ctx->cfg.n_cells,
                                    ctx->stream));
    {
        ProfScope ps(ctx, "bin_sort");
        HIP_TRY(ctx, launch_bin_sort(ctx->st, ctx->stream));
    }
    // Bounds over each cell's populated bins (+ margin), now that the lists say which those are:
    // when the kernel, the environment or a population changed, and every 16th call for the
    // range coagulation itself moves up.  A cell whose bounds still hold costs one look.
    if (ctx->bounds_margin >= 0 &&
        (ctx->bounds_kernel != kernel_type || ctx->bounds_dirty || ++ctx->bounds_age >= 16)) {
        ProfScope ps(ctx, "kernel_bounds");
        HIP_TRY(ctx, launch_kernel_bounds(ctx->st, kernel_type, ctx->bounds_margin, ctx->st.work_list, ctx->stream));
        ctx->bounds_kernel = kernel_type;
        ctx->bounds_dirty = false;
        ctx->bounds_age = 0;
    }
    {
        ProfScope ps(ctx, "coag_sample");
        HIP_TRY(ctx, launch_coag_sample(ctx->st, kernel_type, del_t, seed, step, ctx->stream));
        ctx->sampler_pre_ran = coag_sample_splits_draw(ctx->st, kernel_type);
        ctx->sampler_form = coag_sample_form(ctx->st, kernel_type);
    }
    {
        // close holes in the cells where more than 1/compact_den of the slots are tombstones
        ProfScope ps(ctx, "compact");
        const int den = ctx->cfg.compact_den ? (int)ctx->cfg.compact_den : 2;
        HIP_TRY(ctx, launch_compact(ctx->st, 0, ctx->cfg.n_cells, 1, den, ctx->stream));
    }
    return PMCX_OK;
}

int pmcx_get_coag_counters(pmcx_ctx *ctx, int32_t *n_samp, int32_t *n_coag, int32_t *status)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t b = sizeof(int32_t) * (size_t)ctx->cfg.n_cells;
    if (n_samp) HIP_TRY(ctx, hipMemcpyAsync(n_samp, ctx->st.n_samp, b, hipMemcpyDeviceToHost, ctx->stream));
    if (n_coag) HIP_TRY(ctx, hipMemcpyAsync(n_coag, ctx->st.n_coag, b, hipMemcpyDeviceToHost, ctx->stream));
    if (status) HIP_TRY(ctx, hipMemcpyAsync(status, ctx->st.status, b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_get_coag_totals(pmcx_ctx *ctx, int64_t *n_samp, int64_t *n_coag, int64_t *n_part,
                         int32_t *status_or)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_totals, 0, sizeof(int64_t) * 4, ctx->stream));
    HIP_TRY(ctx, launch_totals(ctx->st, ctx->d_totals, ctx->stream));
    int64_t h[4];
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_totals, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_samp) *n_samp = h[0];
    if (n_coag) *n_coag = h[1];
    if (n_part) *n_part = h[2];
    if (status_or) *status_or = (int32_t)h[3];
    return PMCX_OK;
}

int pmcx_accumulate_totals(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, launch_totals(ctx->st, ctx->d_acc, ctx->stream)); // adds, no host round trip
    return PMCX_OK;
}

int pmcx_read_accumulated(pmcx_ctx *ctx, int64_t *n_samp, int64_t *n_coag, int64_t *n_part,
                          int32_t *status_or, int32_t reset)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int64_t h[4];
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_acc, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof h, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_samp) *n_samp = h[0];
    if (n_coag) *n_coag = h[1];
    if (n_part) *n_part = h[2];
    if (status_or) *status_or = (int32_t)h[3];
    return PMCX_OK;
}

int pmcx_rebalance(pmcx_ctx *ctx, int32_t allow_doubling, int32_t allow_halving, uint64_t seed,
                   uint32_t step)
{
    if (!ctx) return PMCX_EINVAL;
    // with an ideal count of zero every class would be "too large" and halved away
    if (!ctx->n_ideal_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_rebalance: pmcx_set_n_part_ideal must come first");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ProfScope ps(ctx, "rebalance");
    HIP_TRY(ctx, launch_rebalance(ctx->st, allow_doubling, allow_halving, seed, step, ctx->stream));
    return PMCX_OK;
}

namespace {
// The mode parameters of an aero_dist (everything but num_conc) as host arrays in the device's
// layout: parameters of cell c at [c][m] when per_cell (a list's rows go to their cells' places).
int pack_modes(pmcx_ctx *ctx, const pmcx_aero_dist *d, size_t n_rows, const int32_t *row_cell,
               std::vector<double> &h, std::vector<int32_t> &hi, size_t &R)
{
    const int S = ctx->cfg.n_spec, M = d->n_mode;
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (M < 1 || M > 64 || !d->gm_radius || !d->gsd_log10 || !d->vol_frac)
        return fail(ctx, PMCX_EINVAL, "aero_dist: bad n_mode or null array");
    const bool pc = d->per_cell != 0;
    if (pc && (!d->mode_class || !d->mode_source))
        return fail(ctx, PMCX_EINVAL, "aero_dist: per_cell needs mode_class and mode_source");
    if (!n_rows) n_rows = nc;
    R = pc ? nc : 1;
    h.assign(R * (2 * (size_t)M + (size_t)M * S), 0.0);
    hi.assign(R * 2 * (size_t)M, 0);
    double *h_gm = h.data(), *h_gs = h_gm + R * M, *h_vf = h_gs + R * M;
    int32_t *h_mc = hi.data(), *h_ms = h_mc + R * M;
    const size_t rows = pc ? n_rows : 1;
    for (size_t r = 0; r < rows; r++) {
        const size_t c = pc ? (row_cell ? (size_t)row_cell[r] : r) : 0;
        for (int m = 0; m < M; m++) {
            const size_t src = r * M + m, dst = c * M + m;
            if (!(d->gm_radius[src] > 0)) return fail(ctx, PMCX_EINVAL, "aero_dist: gm_radius must be positive");
            h_gm[dst] = det_log(d->gm_radius[src]);     // (not libm's: see pmcx_device_math.h)
            h_gs[dst] = d->gsd_log10[src] * LN_10;
            for (int s = 0; s < S; s++) h_vf[dst * S + s] = d->vol_frac[src * S + s];
            h_mc[dst] = d->mode_class ? d->mode_class[src] : 0;
            h_ms[dst] = d->mode_source ? d->mode_source[src] : 0;
            if (h_mc[dst] < 0 || h_mc[dst] >= ctx->cfg.n_class)
                return fail(ctx, PMCX_EINVAL, "aero_dist: mode_class[%d] out of range", m);
        }
    }
    return PMCX_OK;
}

void modes_view(GenSpecDev &g, int M, size_t R, double *dd, int32_t *di, bool pc)
{
    g.n_mode = M; g.mode_frac = nullptr;
    g.ln_gm_radius = dd; g.ln_gsd = dd + R * M; g.vol_frac = dd + 2 * R * M;
    g.mode_class = di; g.mode_source = di + R * M;
    g.per_cell = pc ? 1 : 0;
}

// uploads one aero_dist into device blocks of the call; returns the device view.  n_rows: cells the
// per-cell arrays of a per_cell dist cover (n_cells, or the entries of a list).  kind: 0 emission,
// 1 background -- a dist whose gm_radius is NULL stands for the resident modes of that kind
// (pmcx_set_aero_scenario_modes) and brings only its num_conc.
int upload_dist(pmcx_ctx *ctx, const pmcx_aero_dist *d, std::vector<void *> &tmp, GenSpecDev &g,
                double *&conc_dev, size_t n_rows = 0, const int32_t *row_cell = nullptr, int kind = -1)
{
    const int M = d->n_mode;
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (!d->num_conc) return fail(ctx, PMCX_EINVAL, "aero_dist: num_conc is NULL");
    if (!d->gm_radius) {
        if (kind < 0 || kind > 1 || !ctx->res_modes[kind].dd)
            return fail(ctx, PMCX_ESTATE, "aero_dist without mode parameters: pmcx_set_aero_scenario_modes "
                                          "has not set the %s modes", kind == 1 ? "background" : "emission");
        g = ctx->res_modes[kind].g;
        if (M != g.n_mode) return fail(ctx, PMCX_EINVAL, "aero_dist: n_mode %d, the resident modes are %d", M, g.n_mode);
    } else {
        std::vector<double> h; std::vector<int32_t> hi; size_t R = 1;
        int rc = pack_modes(ctx, d, n_rows, row_cell, h, hi, R);
        if (rc) return rc;
        double *dd = nullptr; int32_t *di = nullptr;
        HIP_TRY(ctx, tmp_malloc(ctx, (void **)&dd, sizeof(double) * h.size())); tmp.push_back(dd);
        HIP_TRY(ctx, tmp_malloc(ctx, (void **)&di, sizeof(int32_t) * hi.size())); tmp.push_back(di);
        HIP_TRY(ctx, hipMemcpyAsync(dd, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(di, hi.data(), sizeof(int32_t) * hi.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (h, hi go out of scope)
        modes_view(g, M, R, dd, di, d->per_cell != 0);
    }
    HIP_TRY(ctx, tmp_malloc(ctx, (void **)&conc_dev, sizeof(double) * nc * M)); tmp.push_back(conc_dev);
    HIP_TRY(ctx, hipMemcpyAsync(conc_dev, d->num_conc, sizeof(double) * nc * M, hipMemcpyHostToDevice, ctx->stream));
    // (kind >= 0: num_conc is the caller's array, which an asynchronous stage call may leave in flight)
    if (!(ctx->async_stages && kind >= 0)) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}
} // namespace

int pmcx_scenario_update_aero_state(pmcx_ctx *ctx, double dt, double elapsed_time, uint64_t seed,
                                    uint32_t step, const pmcx_aero_dist *emission,
                                    const double *dilution_rate, const pmcx_aero_dist *background,
                                    int32_t *n_emit, int32_t *n_dil_in, int32_t *n_dil_out,
                                    const double *height, const double *old_height)
{
    if (!ctx) return PMCX_EINVAL;
    if (!height || !old_height) height = old_height = nullptr;
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    if (!(dt >= 0)) return fail(ctx, PMCX_EINVAL, "dt must be non-negative");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) tmp_release(ctx, p); };
    GenSpecDev ge{}, gb{};
    double *ce = nullptr, *cbg = nullptr, *d_dil = nullptr, *d_h = nullptr;
    int32_t *d_cnt = nullptr;
    int rc = PMCX_OK;
    if (emission) rc = upload_dist(ctx, emission, tmp, ge, ce, 0, nullptr, 0);
    if (!rc && background) rc = upload_dist(ctx, background, tmp, gb, cbg, 0, nullptr, 1);
    if (rc) { cleanup(); return rc; }
    hipError_t e = hipSuccess;
    if (dilution_rate) {
        e = tmp_malloc(ctx, (void **)&d_dil, sizeof(double) * nc);
        if (e == hipSuccess) { tmp.push_back(d_dil); e = hipMemcpyAsync(d_dil, dilution_rate, sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream); }
    }
    if (e == hipSuccess && height) { // env_state%height, old_env_state%height: the layer's growth
        e = tmp_malloc(ctx, (void **)&d_h, sizeof(double) * 2 * nc);
        if (e == hipSuccess) { tmp.push_back(d_h); e = hipMemcpyAsync(d_h, height, sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream); }
        if (e == hipSuccess) e = hipMemcpyAsync(d_h + nc, old_height, sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) { e = tmp_malloc(ctx, (void **)&d_cnt, sizeof(int32_t) * 3 * nc); if (e == hipSuccess) tmp.push_back(d_cnt); }
    if (e == hipSuccess && !ctx->hold_status) e = hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * nc, ctx->stream);
    if (e == hipSuccess && ctx->auto_grow) { // room for what the adds can bring (an upper bound; the call waits here)
        rc = grow_for_adds(ctx, dt, elapsed_time, emission ? &ge : nullptr, ce, d_dil, background ? &gb : nullptr, cbg, d_h,
                           d_h ? d_h + nc : nullptr);
        if (rc) { cleanup(); return rc; }
    }
    if (e == hipSuccess) {
        ProfScope ps(ctx, "scenario");
        e = launch_scenario_update(ctx->st, dt, elapsed_time, seed, step, emission ? &ge : nullptr, ce,
                                   d_dil, background ? &gb : nullptr, cbg, d_cnt, d_cnt + nc,
                                   d_cnt + 2 * nc, ctx->stream, 0u, d_h, d_h ? d_h + nc : nullptr);
    }
    if (e == hipSuccess && n_emit) e = hipMemcpyAsync(n_emit, d_cnt, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && n_dil_in) e = hipMemcpyAsync(n_dil_in, d_cnt + nc, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && n_dil_out) e = hipMemcpyAsync(n_dil_out, d_cnt + 2 * nc, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = stage_end(ctx);
    cleanup();
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false)); // (the kernel keeps the sort keys current)
    if (e != hipSuccess) return fail(ctx, PMCX_EDEVICE, "pmcx_scenario_update_aero_state: %s", hipGetErrorString(e));
    return PMCX_OK;
}

int pmcx_set_aero_scenario_modes(pmcx_ctx *ctx, int32_t which, const pmcx_aero_dist *dist)
{
    if (!ctx || which < 0 || which > 1) return fail(ctx, PMCX_EINVAL, "pmcx_set_aero_scenario_modes: which is 0 (emission) or 1 (background)");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    auto &r = ctx->res_modes[which];
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (r.dd) { (void)hipFree(r.dd); r.dd = nullptr; }
    if (r.di) { (void)hipFree(r.di); r.di = nullptr; }
    r.g = GenSpecDev{};
    if (!dist) return PMCX_OK; // forgotten
    std::vector<double> h; std::vector<int32_t> hi; size_t R = 1;
    int rc = pack_modes(ctx, dist, 0, nullptr, h, hi, R);
    if (rc) return rc;
    HIP_TRY(ctx, hipMalloc(&r.dd, sizeof(double) * h.size()));
    HIP_TRY(ctx, hipMalloc(&r.di, sizeof(int32_t) * hi.size()));
    HIP_TRY(ctx, hipMemcpyAsync(r.dd, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(r.di, hi.data(), sizeof(int32_t) * hi.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    modes_view(r.g, dist->n_mode, R, (double *)r.dd, (int32_t *)r.di, dist->per_cell != 0);
    return PMCX_OK;
}

int pmcx_seasalt_emissions(pmcx_ctx *ctx, double dt, double elapsed_time, uint64_t seed, uint32_t step,
                           int32_t sea_salt_opt, int32_t i_na, int32_t i_cl,
                           const int32_t *ocean_surface, const double *wind_speed,
                           const double *height, const int32_t *mode_class,
                           const int32_t *mode_source, int32_t *n_emit, double *rate)
{
    if (!ctx || !ocean_surface || !wind_speed || !height)
        return fail(ctx, PMCX_EINVAL, "pmcx_seasalt_emissions: null argument");
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    const int S = ctx->cfg.n_spec;
    if (i_na < 0 || i_na >= S || i_cl < 0 || i_cl >= S || i_na == i_cl || !(dt >= 0))
        return fail(ctx, PMCX_EINVAL, "pmcx_seasalt_emissions: bad species index or dt");
    // the reference's constants (interface/wrf_pmc_driver.F90; default-real literals are the
    // doubles their single-precision values convert to)
    int M = 0;
    double flux[5], gm[5], gsd[5];
    const double f_na = 22.990 / (22.990 + 35.450);                      // driver:1830-1831
    if (sea_salt_opt == 1) {                                             // Gong, MAM3 modes
        const float nf[2] = {73.7095947f, 10.1706810f}, dm[2] = {2.03347408e-07f, 1.09840811e-06f},
                    sg[2] = {2.0f, 2.2f};                                // driver:1819-1822
        M = 2;
        for (int m = 0; m < M; m++) { flux[m] = (double)nf[m]; gm[m] = (double)dm[m] / 2; gsd[m] = std::log10((double)sg[m]); }
    } else if (sea_salt_opt == 2) {                                      // Ovadnevaite 2014
        const float sigma[5] = {1.37f, 1.5f, 1.42f, 1.53f, 1.85f}, cmd[5] = {.018f, .041f, .09f, .23f, .83f},
                    coef[5] = {104.5f, .0442f, 149.6f, 2.96f, .51f}, expon[5] = {.556f, 1.08f, .545f, .79f, .87f};
        const double offset[5] = {1e5, 1e5, 1e5, 1e5, 2e5}, Re = 3.1e6;  // driver:1800-1805, 1909
        M = 5;
        for (int m = 0; m < M; m++) {
            gm[m] = .5 * (double)cmd[m] * 1e-6;
            gsd[m] = std::log10((double)sigma[m]);
            flux[m] = (double)coef[m] * std::pow(Re - offset[m], (double)expon[m]);
        }
    } else {
        return fail(ctx, PMCX_EINVAL, "pmcx_seasalt_emissions: sea_salt_opt must be 1 (Gong) or 2 (Ovadnevaite)");
    }
    const size_t nc = (size_t)ctx->cfg.n_cells;
    std::vector<double> conc(nc * (size_t)M, 0.0), vf((size_t)M * S, 0.0);
    for (size_t c = 0; c < nc; c++) {
        if (!ocean_surface[c]) continue;
        if (!(height[c] > 0) || !(wind_speed[c] >= 0))
            return fail(ctx, PMCX_EINVAL, "pmcx_seasalt_emissions: height must be positive, wind speed non-negative (cell %zu)", c);
        const double scale = sea_salt_opt == 1 ? std::pow(wind_speed[c], (double)3.41f) : 1.0; // driver:1853, 1914
        for (int m = 0; m < M; m++) conc[c * M + m] = flux[m] * scale / height[c];              // driver:1917
    }
    for (int m = 0; m < M; m++) { vf[(size_t)m * S + i_na] = f_na; vf[(size_t)m * S + i_cl] = 1.0 - f_na; }
    if (rate) memcpy(rate, conc.data(), sizeof(double) * conc.size());
    // the reference calls the source on ocean surface cells only (driver:212-223): elsewhere the
    // modes are marked absent, so that nothing -- not even a weight preparation -- happens there
    for (size_t c = 0; c < nc; c++)
        if (!ocean_surface[c])
            for (int m = 0; m < M; m++) conc[c * M + m] = -1.0;
    pmcx_aero_dist d{};
    d.n_mode = M; d.num_conc = conc.data(); d.gm_radius = gm; d.gsd_log10 = gsd; d.vol_frac = vf.data();
    d.mode_class = mode_class; d.mode_source = mode_source;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) tmp_release(ctx, p); };
    GenSpecDev ge{};
    double *ce = nullptr;
    int32_t *d_cnt = nullptr;
    int rc = upload_dist(ctx, &d, tmp, ge, ce);
    if (rc) { cleanup(); return rc; }
    hipError_t e = tmp_malloc(ctx, (void **)&d_cnt, sizeof(int32_t) * nc);
    if (e == hipSuccess) tmp.push_back(d_cnt);
    if (e == hipSuccess && !ctx->hold_status) e = hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * nc, ctx->stream);
    if (e == hipSuccess && ctx->auto_grow) {
        rc = grow_for_adds(ctx, dt, elapsed_time, &ge, ce, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc) { cleanup(); return rc; }
    }
    if (e == hipSuccess) {
        ProfScope ps(ctx, "scenario");
        e = launch_scenario_update(ctx->st, dt, elapsed_time, seed, step, &ge, ce, nullptr, nullptr, nullptr,
                                   d_cnt, nullptr, nullptr, ctx->stream, 2u);
    }
    if (e == hipSuccess && n_emit) e = hipMemcpyAsync(n_emit, d_cnt, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false));
    if (e != hipSuccess) return fail(ctx, PMCX_EDEVICE, "pmcx_seasalt_emissions: %s", hipGetErrorString(e));
    return PMCX_OK;
}

int pmcx_dry_deposition(pmcx_ctx *ctx, double dt, uint64_t seed, uint32_t step,
                        const int32_t *surface, const double *aer_res_a, const double *ustar,
                        const double *alpha, const double *big_a, const double *height,
                        const double *inv_density, int32_t *n_removed, int32_t *unrealistic)
{
    if (!ctx || !aer_res_a || !ustar || !alpha || !big_a || !height || !inv_density)
        return fail(ctx, PMCX_EINVAL, "pmcx_dry_deposition: null argument");
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    if (!(dt >= 0)) return fail(ctx, PMCX_EINVAL, "dt must be non-negative");
    const size_t nc = (size_t)ctx->cfg.n_cells;
    for (size_t k = 0; k < nc; k++)
        if ((!surface || surface[k]) && (!(height[k] > 0) || !(ustar[k] > 0) || !(big_a[k] > 0)))
            return fail(ctx, PMCX_EINVAL, "pmcx_dry_deposition: height, ustar and big_a must be positive (cell %zu)", k);
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    double *d6 = nullptr;
    int32_t *d_i = nullptr;
    hipError_t e = tmp_malloc(ctx, (void **)&d6, sizeof(double) * 6 * nc);
    if (e == hipSuccess) e = tmp_malloc(ctx, (void **)&d_i, sizeof(int32_t) * 3 * nc);
    const double *src[6] = {aer_res_a, ustar, alpha, big_a, height, inv_density};
    for (int k = 0; k < 6 && e == hipSuccess; k++)
        e = hipMemcpyAsync(d6 + k * nc, src[k], sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && surface)
        e = hipMemcpyAsync(d_i + 2 * nc, surface, sizeof(int32_t) * nc, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        ProfScope ps(ctx, "dry_deposition");
        e = launch_dry_deposition(ctx->st, dt, seed, step, surface ? d_i + 2 * nc : nullptr, d6, d_i,
                                  d_i + nc, ctx->stream);
    }
    if (e == hipSuccess && n_removed) e = hipMemcpyAsync(n_removed, d_i, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && unrealistic) e = hipMemcpyAsync(unrealistic, d_i + nc, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = stage_end(ctx);
    if (d6) tmp_release(ctx, d6);
    if (d_i) tmp_release(ctx, d_i);
    if (e != hipSuccess) return fail(ctx, PMCX_EDEVICE, "pmcx_dry_deposition: %s", hipGetErrorString(e));
    return PMCX_OK;
}

int pmcx_transport(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                   int32_t periodic_y, const double *prob_h, const double *prob_v,
                   const double *cell_volume, uint64_t seed, uint32_t step, int32_t *n_moved,
                   int32_t *n_arrived)
{
    if (!ctx || !cell_volume || (!prob_h) != (!prob_v))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport: null argument");
    const bool built = !prob_h; // the probabilities pmcx_transport_probs left on the device
    if (built && (ctx->probs_nx != nx || ctx->probs_nz != nz || ctx->probs_ny != ny))
        return fail(ctx, PMCX_ESTATE, "pmcx_transport: no probabilities were built for this grid "
                                      "(pmcx_transport_probs comes first, once per transport)");
    const pmcx_config &cf = ctx->cfg;
    if (nx < 1 || nz < 1 || ny < 1 || nz > 128 || (int64_t)nx * nz * ny != cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport: nx*nz*ny must equal n_cells (nz <= 128)");
    if (ctx->st.trans_cut_x)
        return fail(ctx, PMCX_ESTATE, "pmcx_transport: the block is one of a domain cut in x (pmcx_set_transport_cuts): "
                                      "its transport goes through pmcx_transport_begin ... _finish");
    if (ctx->st.trans_lateral && (periodic_x || periodic_y || nx < 3 || ny < 3 || ctx->st.trans_j0 != 0 ||
                                  (ctx->st.trans_gny != 0 && ctx->st.trans_gny != ny)))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport: open lateral boundaries (pmcx_set_transport_lateral) "
                                      "need a non-periodic block of at least 3 x 3 columns that is the whole "
                                      "domain (a slab goes through pmcx_transport_begin)");
    if (transport_lds_bytes(ctx->st, nz) > 160 * 1024)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport: n_class x (4+nz) destination tables exceed LDS");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class;
    if (!ctx->d_outbox) {
        int rc = dev_alloc(ctx, (int64_t **)&ctx->d_outbox, nc * (size_t)cf.cap, false);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_out, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_arr, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_dest_mask, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_exp, nc);
        if (rc) return rc;
    }
    const size_t bh = sizeof(double) * nc * C * 4, bv = sizeof(double) * nc * C * (size_t)nz,
                 bc = sizeof(double) * nc;
    int rc = ensure_stage(ctx, bh + bv + bc + 64);
    if (rc) return rc;
    double *d_ph = (double *)ctx->stage, *d_pv = d_ph + nc * C * 4, *d_cv = d_pv + nc * C * (size_t)nz;
    if (built) {
        d_ph = ctx->d_prob_h; d_pv = ctx->d_prob_v;
        ctx->probs_nx = 0; // (they describe the populations as they were: used once)
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(d_ph, prob_h, bh, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pv, prob_v, bv, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_cv, cell_volume, bc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * nc, ctx->stream));
    if (!ctx->auto_grow) {
        ProfScope ps(ctx, "transport");
        HIP_TRY(ctx, launch_transport(ctx->st, nx, nz, ny, periodic_x, periodic_y, d_ph, d_pv, d_cv,
                                      ctx->d_outbox, ctx->d_n_out, ctx->d_n_arr, ctx->d_n_exp,
                                      ctx->d_dest_mask, seed, step, ctx->stream));
    } else {
        // pmcx_set_auto_grow: the arrivals every cell is about to take are counted by its sources
        // before anything is written to it -- a cell that would not hold them gets its slots first
        {
            ProfScope ps(ctx, "transport");
            rc = decide_with_room(ctx, [&]() {
                return launch_transport_begin(ctx->st, nx, nz, ny, periodic_x, periodic_y, d_ph, d_pv, d_cv,
                                              ctx->d_outbox, ctx->d_n_out, ctx->d_n_arr, ctx->d_n_exp,
                                              ctx->d_dest_mask, seed, step, ctx->stream);
            });
            if (rc) return rc;
        }
        // (the cell volumes live in the staging area, which growing does not touch)
        rc = grow_if_needed(ctx, ctx->d_n_exp);
        if (rc) return rc;
        ProfScope ps(ctx, "transport");
        HIP_TRY(ctx, launch_transport_finish(ctx->st, nx, nz, ny, periodic_x, periodic_y, d_cv, ctx->d_outbox,
                                             ctx->d_n_out, ctx->d_n_arr, ctx->d_n_exp, ctx->d_dest_mask, 0, ny,
                                             ctx->stream));
    }
    if (n_moved) HIP_TRY(ctx, hipMemcpyAsync(n_moved, ctx->d_n_out, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    if (n_arrived) HIP_TRY(ctx, hipMemcpyAsync(n_arrived, ctx->d_n_arr, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stage_end(ctx));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false)); // (the kernels keep the sort keys current)
    return PMCX_OK;
}

int pmcx_transport_probs(pmcx_ctx *ctx, const pmcx_trans_fields *f, double *prob_h, double *prob_v,
                         int32_t *n_substeps, int32_t *bad, int32_t *n_class_part)
{
    if (!ctx || !f || !f->alt || !f->xkhh || !f->u_flux || !f->v_flux || !f->w_flux || !f->z ||
        !f->z_at_w || !f->exch_h)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: null argument");
    const pmcx_config &cf = ctx->cfg;
    const int nx = f->nx, nz = f->nz, ny = f->ny;
    if (nx < 1 || nz < 2 || ny < 1 || nz > 128 || (int64_t)nx * nz * ny != cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: nx*nz*ny must equal n_cells, 2 <= nz <= 128");
    if (!(f->dt > 0) || !(f->dx > 0) || !(f->dy > 0))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: dt, dx, dy must be positive");
    if (trans_probs_lds_bytes(ctx->st, nz) > 160 * 1024)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: the column matrices (2 nz^2 doubles) exceed LDS");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class, col = (size_t)nx * ny;
    if (!ctx->d_prob_h || ctx->probs_nz != nz) {
        if (ctx->d_prob_h) return fail(ctx, PMCX_ESTATE, "pmcx_transport_probs: nz differs from the first call's");
        int rc = dev_alloc(ctx, &ctx->d_prob_h, nc * C * 4);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_prob_v, nc * C * (size_t)nz);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_prob_i, nc + nc * C + nc + 1);
        if (rc) return rc;
        ctx->probs_nz = nz;
    }
    ctx->probs_nx = 0;
    // the fields, single precision as in WRF's grid%
    const size_t n_halo = (size_t)(ny + 2) * nz * (nx + 2), n_u = C * ny * nz * (size_t)(nx + 1),
                 n_v = C * (size_t)(ny + 1) * nz * nx, n_w = C * ny * (size_t)(nz + 1) * nx,
                 n_z = (size_t)ny * nz * nx, n_zw = (size_t)ny * (nz + 1) * nx;
    const size_t counts[8] = {n_halo, n_halo, n_u, n_v, n_w, n_z, n_zw, n_zw};
    const float *src[8] = {f->alt, f->xkhh, f->u_flux, f->v_flux, f->w_flux, f->z, f->z_at_w, f->exch_h};
    size_t total = 0;
    for (size_t n : counts) total += (n + 3) & ~(size_t)3;
    int rc = ensure_stage(ctx, sizeof(float) * total + 64);
    if (rc) return rc;
    const float *dev[8];
    float *at = (float *)ctx->stage;
    for (int k = 0; k < 8; k++) {
        HIP_TRY(ctx, hipMemcpyAsync(at, src[k], sizeof(float) * counts[k], hipMemcpyHostToDevice, ctx->stream));
        dev[k] = at;
        at += (counts[k] + 3) & ~(size_t)3;
    }
    TransFieldsArgs a{};
    a.nx = nx; a.nz = nz; a.ny = ny; a.do_vertmix = f->do_vertmix; a.all_columns = f->all_columns;
    a.i_offset = f->i_offset; a.j_offset = f->j_offset; a.global_nx = f->global_nx; a.global_ny = f->global_ny;
    a.dt = f->dt; a.dx = f->dx; a.dy = f->dy;
    a.alt = dev[0]; a.xkhh = dev[1]; a.u_flux = dev[2]; a.v_flux = dev[3]; a.w_flux = dev[4];
    a.z = dev[5]; a.z_at_w = dev[6]; a.exch_h = dev[7];
    a.prob_h = ctx->d_prob_h; a.prob_v = ctx->d_prob_v;
    a.bad = ctx->d_prob_i; a.n_class_part = ctx->d_prob_i + nc; a.n_substeps = ctx->d_prob_i + nc + nc * C;
    a.clamp_flag = ctx->d_prob_i + nc + nc * C + nc;
    HIP_TRY(ctx, hipMemsetAsync(a.clamp_flag, 0, sizeof(int32_t), ctx->stream));
    {
        ProfScope ps(ctx, "transport_probs");
        HIP_TRY(ctx, launch_transport_probs(ctx->st, a, ctx->stream));
    }
    int32_t clamped = 0;
    if (ctx->async_stages) // (reported by the next pmcx_synchronize)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, a.clamp_flag, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    else
        HIP_TRY(ctx, hipMemcpyAsync(&clamped, a.clamp_flag, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (prob_h) HIP_TRY(ctx, hipMemcpyAsync(prob_h, ctx->d_prob_h, sizeof(double) * nc * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (prob_v) HIP_TRY(ctx, hipMemcpyAsync(prob_v, ctx->d_prob_v, sizeof(double) * nc * C * (size_t)nz, hipMemcpyDeviceToHost, ctx->stream));
    if (bad) HIP_TRY(ctx, hipMemcpyAsync(bad, a.bad, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    if (n_class_part) HIP_TRY(ctx, hipMemcpyAsync(n_class_part, a.n_class_part, sizeof(int32_t) * nc * C, hipMemcpyDeviceToHost, ctx->stream));
    if (n_substeps) HIP_TRY(ctx, hipMemcpyAsync(n_substeps, a.n_substeps, sizeof(int32_t) * col, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stage_end(ctx));
    if (clamped)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_probs: a column asks for more than 10^4 vertical "
                                      "sub-steps (z(2) - z(1) too small or exch_h too large; bit 16 of bad[]): "
                                      "the probabilities were not built");
    ctx->probs_nx = nx; ctx->probs_ny = ny;
    ctx->cls_fresh = cf.n_class > 1;
    return PMCX_OK;
}

int pmcx_boundary_copy(pmcx_ctx *ctx, int32_t n_pairs, const int32_t *from_cell, const int32_t *to_cell)
{
    if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!from_cell || !to_cell)))
        return fail(ctx, PMCX_EINVAL, "pmcx_boundary_copy: bad argument");
    if (n_pairs == 0) return PMCX_OK;
    const int nc = ctx->cfg.n_cells;
    std::vector<unsigned char> is_to((size_t)nc, 0);
    for (int q = 0; q < n_pairs; q++) {
        if (from_cell[q] < 0 || from_cell[q] >= nc || to_cell[q] < 0 || to_cell[q] >= nc || from_cell[q] == to_cell[q])
            return fail(ctx, PMCX_EINVAL, "pmcx_boundary_copy: pair %d names a cell outside the block (or itself)", q);
        if (is_to[(size_t)to_cell[q]]++) return fail(ctx, PMCX_EINVAL, "pmcx_boundary_copy: cell %d is filled twice", to_cell[q]);
    }
    for (int q = 0; q < n_pairs; q++)
        if (is_to[(size_t)from_cell[q]])
            return fail(ctx, PMCX_EINVAL, "pmcx_boundary_copy: cell %d is both copied from and filled", from_cell[q]);
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_stage(ctx, sizeof(int32_t) * 2 * (size_t)n_pairs + 64);
    if (rc) return rc;
    int32_t *d_from = (int32_t *)ctx->stage, *d_to = d_from + n_pairs;
    HIP_TRY(ctx, hipMemcpyAsync(d_from, from_cell, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_to, to_cell, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "boundary");
        HIP_TRY(ctx, launch_boundary_copy(ctx->st, n_pairs, d_from, d_to, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->bounds_dirty = true;
    return PMCX_OK;
}

int pmcx_boundary_resample(pmcx_ctx *ctx, int32_t n_list, const int32_t *cell, const int32_t *throw_away,
                           const pmcx_aero_dist *background, const double *density, double create_time,
                           uint64_t seed, uint32_t step, int32_t *n_added, int32_t *n_removed)
{
    if (!ctx || n_list < 0 || (n_list > 0 && (!cell || !background || !density)))
        return fail(ctx, PMCX_EINVAL, "pmcx_boundary_resample: bad argument");
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    if (n_list == 0) return PMCX_OK;
    if (int grc = grow_if_needed(ctx, nullptr)) return grc; // (pmcx_set_auto_grow: room for the refill)
    const size_t nc = (size_t)ctx->cfg.n_cells;
    const int M = background->n_mode;
    if (M < 1 || M > 64 || !background->num_conc) return fail(ctx, PMCX_EINVAL, "pmcx_boundary_resample: bad background");
    // per-cell arrays over the whole block (the kernel's wavefronts leave cells without action alone)
    std::vector<int32_t> act(nc, 0);
    std::vector<double> conc(nc * (size_t)M, 0.0), dens(nc, 0.0);
    for (int q = 0; q < n_list; q++) {
        const int c = cell[q];
        if (c < 0 || (size_t)c >= nc || act[(size_t)c])
            return fail(ctx, PMCX_EINVAL, "pmcx_boundary_resample: entry %d names a cell outside the block or twice", q);
        if (!(density[q] > 0)) return fail(ctx, PMCX_EINVAL, "pmcx_boundary_resample: density must be positive");
        act[(size_t)c] = (throw_away && throw_away[q]) ? 2 : 1;
        dens[(size_t)c] = density[q];
        for (int m = 0; m < M; m++) conc[(size_t)c * M + m] = background->num_conc[(size_t)q * M + m];
    }
    pmcx_aero_dist d = *background;
    d.num_conc = conc.data();
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) tmp_release(ctx, p); };
    GenSpecDev gb{};
    double *cbg = nullptr, *d_dens = nullptr;
    int32_t *d_i = nullptr;
    int rc = upload_dist(ctx, &d, tmp, gb, cbg, (size_t)n_list, cell); // (per_cell: rows follow the list)
    if (rc) { cleanup(); return rc; }
    hipError_t e = tmp_malloc(ctx, (void **)&d_dens, sizeof(double) * nc);
    if (e == hipSuccess) tmp.push_back(d_dens);
    if (e == hipSuccess) e = tmp_malloc(ctx, (void **)&d_i, sizeof(int32_t) * 3 * nc);
    if (e == hipSuccess) tmp.push_back(d_i);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dens, dens.data(), sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_i, act.data(), sizeof(int32_t) * nc, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_i + nc, 0, sizeof(int32_t) * 2 * nc, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * nc, ctx->stream);
    if (e == hipSuccess) {
        ProfScope ps(ctx, "boundary");
        e = launch_boundary_resample(ctx->st, d_i, gb, cbg, d_dens, create_time, seed, step, d_i + nc,
                                     d_i + 2 * nc, ctx->stream);
    }
    std::vector<int32_t> h_cnt(2 * nc);
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt.data(), d_i + nc, sizeof(int32_t) * 2 * nc, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return fail(ctx, PMCX_EDEVICE, "pmcx_boundary_resample: %s", hipGetErrorString(e));
    for (int q = 0; q < n_list; q++) {
        if (n_added) n_added[q] = h_cnt[(size_t)cell[q]];
        if (n_removed) n_removed[q] = h_cnt[nc + (size_t)cell[q]];
    }
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false));
    return PMCX_OK;
}

int pmcx_transport_preweight(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                             int32_t periodic_y, const double *prob_h, const double *prob_v,
                             const double *cell_volume, uint64_t seed, uint32_t step,
                             double *expected_num_conc, int32_t *n_removed)
{
    if (!ctx || !cell_volume || (!prob_h) != (!prob_v))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_preweight: null argument");
    const bool counted = ctx->cls_fresh; // (pmcx_transport_probs was the call in front of this one)
    const bool built = !prob_h;
    if (built && (ctx->probs_nx != nx || ctx->probs_nz != nz || ctx->probs_ny != ny))
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight: no probabilities were built for this grid");
    if (!ctx->n_ideal_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_n_part_ideal must come first");
    const pmcx_config &cf = ctx->cfg;
    if (nx < 1 || nz < 1 || ny < 1 || nz > 128 || (int64_t)nx * nz * ny != cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_preweight: nx*nz*ny must equal n_cells (nz <= 128)");
    if (ctx->st.trans_cut_x)
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight: the block is one of a domain cut in x: its preweight "
                                      "goes through pmcx_transport_preweight_begin, _x_terms, _finish");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class;
    if (!ctx->d_preweight) {
        int rc = dev_alloc(ctx, &ctx->d_preweight, 2 * nc * C);
        if (rc) return rc;
    }
    const size_t bh = sizeof(double) * nc * C * 4, bv = sizeof(double) * nc * C * (size_t)nz,
                 bc = sizeof(double) * nc;
    int rc = ensure_stage(ctx, bh + bv + bc + sizeof(int32_t) * nc + 64);
    if (rc) return rc;
    double *d_ph = (double *)ctx->stage, *d_pv = d_ph + nc * C * 4, *d_cv = d_pv + nc * C * (size_t)nz;
    int32_t *d_nr = (int32_t *)(d_cv + nc);
    if (built) { d_ph = ctx->d_prob_h; d_pv = ctx->d_prob_v; } // (left in place for the transport)
    else {
        HIP_TRY(ctx, hipMemcpyAsync(d_ph, prob_h, bh, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pv, prob_v, bv, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_cv, cell_volume, bc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_nr, 0, sizeof(int32_t) * nc, ctx->stream));
    {
        ProfScope ps(ctx, "transport_preweight");
        ctx->st.cls_fresh = counted ? 1 : 0;
        HIP_TRY(ctx, launch_transport_preweight(ctx->st, nx, nz, ny, periodic_x && periodic_y, d_ph, d_pv,
                                                d_cv, ctx->d_preweight, ctx->d_preweight + nc * C, d_nr,
                                                seed, step, ctx->stream));
    }
    if (expected_num_conc)
        HIP_TRY(ctx, hipMemcpyAsync(expected_num_conc, ctx->d_preweight + nc * C, sizeof(double) * nc * C,
                                    hipMemcpyDeviceToHost, ctx->stream));
    if (n_removed) HIP_TRY(ctx, hipMemcpyAsync(n_removed, d_nr, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stage_end(ctx));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, cf.n_cells, false)); // (removals keep the sort keys current)
    ctx->bounds_dirty = true;
    return PMCX_OK;
}

// ---- the preweight of a slab: the sums over the cut rows come from the neighbouring ranks ----

static int transport_buffers(pmcx_ctx *ctx);

int pmcx_transport_preweight_begin(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                                   int32_t periodic_y, int32_t j_global0, int32_t global_ny,
                                   const double *prob_h, const double *prob_v, const double *cell_volume,
                                   double *edge_terms)
{
    if (!ctx || !cell_volume || !edge_terms || (!prob_h) != (!prob_v))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_preweight_begin: null argument");
    const bool counted = ctx->cls_fresh; // (pmcx_transport_probs was the call in front of this one)
    const pmcx_config &cf = ctx->cfg;
    if (nx < 1 || nz < 1 || ny < 3 || nz > 128 || (int64_t)nx * nz * ny != cf.n_cells || global_ny < ny - 2)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_preweight_begin: nx*nz*ny must equal n_cells, ny >= 3 "
                                      "(two ghost rows), nz <= 128");
    if (!ctx->n_ideal_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_n_part_ideal must come first");
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class, row = (size_t)nx * nz;
    int rc = transport_buffers(ctx);
    if (rc) return rc;
    if (!ctx->d_preweight) { rc = dev_alloc(ctx, &ctx->d_preweight, 2 * nc * C); if (rc) return rc; }
    // ready terms out and in: two rows each, and (a block cut in x as well) two columns each
    const size_t col = (size_t)ny * nz, pw_need = 4 * (row + col) * C;
    if (!ctx->d_pw_edge || ctx->pw_edge_len < pw_need) {
        rc = dev_alloc(ctx, &ctx->d_pw_edge, pw_need); // (a smaller one, if any, stays allocated until destroy)
        if (rc) return rc;
        ctx->pw_edge_len = pw_need;
    }
    if (ctx->st.trans_cut_x && (nx < 3 || ctx->st.trans_gnx < nx - 2))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_preweight_begin: a block cut in x has two ghost columns (nx >= 3)");
    if (prob_h) { // kept on the device for _finish and for the transport that follows
        if (!ctx->d_prob_h) {
            rc = dev_alloc(ctx, &ctx->d_prob_h, nc * C * 4);
            if (!rc) rc = dev_alloc(ctx, &ctx->d_prob_v, nc * C * (size_t)nz);
            if (!rc) rc = dev_alloc(ctx, &ctx->d_prob_i, nc + nc * C + nc + 1);
            if (rc) return rc;
            ctx->probs_nz = nz;
        } else if (ctx->probs_nz != nz) {
            return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight_begin: nz differs from the first call's");
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_prob_h, prob_h, sizeof(double) * nc * C * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_prob_v, prob_v, sizeof(double) * nc * C * (size_t)nz, hipMemcpyHostToDevice, ctx->stream));
        ctx->probs_nx = nx; ctx->probs_ny = ny;
    } else if (ctx->probs_nx != nx || ctx->probs_nz != nz || ctx->probs_ny != ny) {
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight_begin: no probabilities were built for this grid");
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_halo_cv, cell_volume, sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream));
    const int ideal = periodic_x && periodic_y;
    {
        ProfScope ps(ctx, "transport_preweight");
        ctx->st.cls_fresh = counted ? 1 : 0;
        HIP_TRY(ctx, launch_preweight_slab_begin(ctx->st, nx, nz, ny, ideal, j_global0, global_ny, ctx->d_prob_h,
                                                 ctx->d_prob_v, ctx->d_halo_cv, ctx->d_preweight, ctx->d_pw_edge,
                                                 ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(edge_terms, ctx->d_pw_edge, sizeof(double) * (2 * row + (ctx->st.trans_cut_x ? 2 * col : 0)) * C,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pw_x_set = false;
    ctx->pw_nx = nx; ctx->pw_nz = nz; ctx->pw_ny = ny; ctx->pw_ideal = ideal;
    ctx->pw_jg0 = j_global0; ctx->pw_gny = global_ny;
    return PMCX_OK;
}

int pmcx_transport_preweight_x_terms(pmcx_ctx *ctx, const double *terms_from_west, const double *terms_from_east,
                                     int32_t west_is_edge, int32_t east_is_edge)
{
    if (!ctx) return PMCX_EINVAL;
    if (!ctx->pw_nx || !ctx->st.trans_cut_x)
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight_x_terms: no preweight of a block cut in x is open");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t C = (size_t)ctx->cfg.n_class, row = (size_t)ctx->pw_nx * ctx->pw_nz, col = (size_t)ctx->pw_ny * ctx->pw_nz;
    double *g_w = ctx->d_pw_edge + (4 * row + 2 * col) * C, *g_e = g_w + col * C;
    if (terms_from_west) HIP_TRY(ctx, hipMemcpyAsync(g_w, terms_from_west, sizeof(double) * col * C, hipMemcpyHostToDevice, ctx->stream));
    if (terms_from_east) HIP_TRY(ctx, hipMemcpyAsync(g_e, terms_from_east, sizeof(double) * col * C, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pw_x_set = true;
    ctx->pw_gin_w = terms_from_west ? g_w : nullptr; ctx->pw_gin_e = terms_from_east ? g_e : nullptr;
    ctx->pw_w_edge = west_is_edge; ctx->pw_e_edge = east_is_edge;
    return PMCX_OK;
}

int pmcx_transport_preweight_finish(pmcx_ctx *ctx, const double *terms_from_below, const double *terms_from_above,
                                    int32_t below_is_edge, int32_t above_is_edge, uint64_t seed, uint32_t step,
                                    double *expected_num_conc, int32_t *n_removed)
{
    if (!ctx) return PMCX_EINVAL;
    if (!ctx->pw_nx) return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight_finish: no preweight is open");
    const pmcx_config &cf = ctx->cfg;
    const int nx = ctx->pw_nx, nz = ctx->pw_nz, ny = ctx->pw_ny;
    if (ctx->st.trans_cut_x && !ctx->pw_x_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_preweight_finish: the block is cut in x: "
                                      "pmcx_transport_preweight_x_terms comes first");
    ctx->pw_nx = 0;
    HIP_TRY(ctx, hipSetDevice(cf.device));
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class, row = (size_t)nx * nz, col = (size_t)ny * nz;
    // (layout of d_pw_edge: out rows [2 row], out columns [2 col], in rows [2 row], in columns [2 col])
    double *g_lo = ctx->d_pw_edge + (2 * row + 2 * col) * C, *g_hi = g_lo + row * C;
    if (terms_from_below) HIP_TRY(ctx, hipMemcpyAsync(g_lo, terms_from_below, sizeof(double) * row * C, hipMemcpyHostToDevice, ctx->stream));
    if (terms_from_above) HIP_TRY(ctx, hipMemcpyAsync(g_hi, terms_from_above, sizeof(double) * row * C, hipMemcpyHostToDevice, ctx->stream));
    int rc = ensure_stage(ctx, sizeof(int32_t) * nc + 64);
    if (rc) return rc;
    int32_t *d_nr = (int32_t *)ctx->stage;
    HIP_TRY(ctx, hipMemsetAsync(d_nr, 0, sizeof(int32_t) * nc, ctx->stream));
    {
        ProfScope ps(ctx, "transport_preweight");
        HIP_TRY(ctx, launch_preweight_slab_finish(ctx->st, nx, nz, ny, ctx->pw_ideal, ctx->pw_jg0, ctx->pw_gny,
                                                  below_is_edge, above_is_edge, ctx->d_prob_h, ctx->d_prob_v,
                                                  ctx->d_halo_cv, ctx->d_preweight, ctx->d_preweight + nc * C,
                                                  terms_from_below ? g_lo : nullptr,
                                                  terms_from_above ? g_hi : nullptr, d_nr, seed, step, ctx->stream,
                                                  ctx->pw_w_edge, ctx->pw_e_edge,
                                                  ctx->st.trans_cut_x ? ctx->pw_gin_w : nullptr,
                                                  ctx->st.trans_cut_x ? ctx->pw_gin_e : nullptr));
    }
    if (expected_num_conc)
        HIP_TRY(ctx, hipMemcpyAsync(expected_num_conc, ctx->d_preweight + nc * C, sizeof(double) * nc * C,
                                    hipMemcpyDeviceToHost, ctx->stream));
    if (n_removed) HIP_TRY(ctx, hipMemcpyAsync(n_removed, d_nr, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, cf.n_cells, false));
    return PMCX_OK;
}

// ---- transport of a slab whose first and last rows are ghosts of the neighbouring ranks ----

static int transport_buffers(pmcx_ctx *ctx)
{
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (!ctx->d_outbox) {
        int rc = dev_alloc(ctx, (int64_t **)&ctx->d_outbox, nc * (size_t)ctx->cfg.cap, false);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_out, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_arr, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_dest_mask, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_n_exp, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_halo_cv, nc);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_halo_bad, (size_t)1);
        if (rc) return rc;
    }
    return PMCX_OK;
}

int pmcx_device_alloc(pmcx_ctx *ctx, int64_t n_bytes, void **dev_ptr)
{
    if (!ctx || !dev_ptr || n_bytes < 0) return fail(ctx, PMCX_EINVAL, "pmcx_device_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    *dev_ptr = nullptr;
    HIP_TRY(ctx, hipMalloc(dev_ptr, (size_t)(n_bytes > 0 ? n_bytes : 1)));
    return PMCX_OK;
}
int pmcx_device_free(pmcx_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (dev_ptr) HIP_TRY(ctx, hipFree(dev_ptr));
    return PMCX_OK;
}
int pmcx_device_copy(pmcx_ctx *ctx, void *dst, const void *src, int64_t n_bytes, int32_t kind)
{
    if (!ctx || (n_bytes > 0 && (!dst || !src)) || kind < 0 || kind > 2)
        return fail(ctx, PMCX_EINVAL, "pmcx_device_copy: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : kind == 1 ? hipMemcpyHostToDevice
                                                                          : hipMemcpyDeviceToHost;
    if (n_bytes > 0) HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)n_bytes, k, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_host_alloc(pmcx_ctx *ctx, int64_t n_bytes, void **host_ptr)
{
    if (!ctx || !host_ptr || n_bytes < 1) return fail(ctx, PMCX_EINVAL, "pmcx_host_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    *host_ptr = nullptr;
    const hipError_t e = hipHostMalloc(host_ptr, (size_t)n_bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, PMCX_ENOMEM, "pmcx_host_alloc(%lld bytes): %s", (long long)n_bytes, hipGetErrorString(e));
    return PMCX_OK;
}
int pmcx_host_free(pmcx_ctx *ctx, void *host_ptr)
{
    if (!ctx) return PMCX_EINVAL;
    if (host_ptr) HIP_TRY(ctx, hipHostFree(host_ptr));
    return PMCX_OK;
}

int pmcx_transport_begin(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                         const double *prob_h, const double *prob_v, const double *cell_volume,
                         uint64_t seed, uint32_t step)
{
    if (!ctx || !cell_volume || (!prob_h) != (!prob_v))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_begin: null argument");
    const bool built = !prob_h;
    if (built && (ctx->probs_nx != nx || ctx->probs_nz != nz || ctx->probs_ny != ny))
        return fail(ctx, PMCX_ESTATE, "pmcx_transport_begin: no probabilities were built for this grid "
                                      "(pmcx_transport_probs comes first, once per transport)");
    const pmcx_config &cf = ctx->cfg;
    if (nx < 1 || nz < 1 || ny < 3 || nz > 128 || (int64_t)nx * nz * ny != cf.n_cells)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_begin: nx*nz*ny must equal n_cells, ny >= 3 "
                                      "(two ghost rows), nz <= 128");
    if (transport_lds_bytes(ctx->st, nz) > 160 * 1024)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_begin: destination tables exceed LDS");
    if (ctx->st.trans_lateral && (periodic_x || nx < 3 || ctx->st.trans_gny < 3))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_begin: open lateral boundaries need a non-periodic domain "
                                      "and the slab's place in it (pmcx_set_transport_lateral)");
    if (ctx->st.trans_cut_x && (nx < 3 || ctx->st.trans_gnx < nx - 2))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_begin: a block cut in x has two ghost columns (nx >= 3) and "
                                      "knows the domain's columns (pmcx_set_transport_cuts)");
    if (ctx->st.trans_cut_x) periodic_x = 0; // (the ghost columns stand for whatever lies beyond)
    HIP_TRY(ctx, hipSetDevice(cf.device));
    int rc = transport_buffers(ctx);
    if (rc) return rc;
    const size_t nc = (size_t)cf.n_cells, C = (size_t)cf.n_class;
    const size_t row = std::max((size_t)nx * nz, ctx->st.trans_cut_x ? (size_t)ny * nz : (size_t)0); // the longest strip
    if (!ctx->d_halo || ctx->halo_row_cells < (int)row) { // grown only when the strips get longer
        if (ctx->d_halo) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (auto it = ctx->allocs.begin(); it != ctx->allocs.end(); ++it)
                if (*it == (void *)ctx->d_halo) { ctx->allocs.erase(it); break; }
            HIP_TRY(ctx, hipFree(ctx->d_halo));
            ctx->bytes -= (int64_t)sizeof(int32_t) * 12 * ((int64_t)ctx->halo_row_cells + 1);
            ctx->d_halo = nullptr;
        }
        rc = dev_alloc(ctx, &ctx->d_halo, 12 * (row + 1)); // counts + offsets of 4 export sides, offsets of 4 imports
        if (rc) return rc;
        ctx->halo_row_cells = (int)row;
    }
    const size_t bh = sizeof(double) * nc * C * 4, bv = sizeof(double) * nc * C * (size_t)nz,
                 bc = sizeof(double) * nc;
    rc = ensure_stage(ctx, bh + bv + bc + 64);
    if (rc) return rc;
    double *d_ph = (double *)ctx->stage, *d_pv = d_ph + nc * C * 4, *d_cv = ctx->d_halo_cv;
    if (built) {
        d_ph = ctx->d_prob_h; d_pv = ctx->d_prob_v;
        ctx->probs_nx = 0;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(d_ph, prob_h, bh, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pv, prob_v, bv, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_cv, cell_volume, bc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * nc, ctx->stream));
    ctx->halo_nx = nx; ctx->halo_nz = nz; ctx->halo_ny = ny; ctx->halo_px = periodic_x;
    ctx->halo_total[0] = ctx->halo_total[1] = ctx->halo_total[2] = ctx->halo_total[3] = -1;
    ProfScope ps(ctx, "transport");
    rc = decide_with_room(ctx, [&]() {
        return launch_transport_begin(ctx->st, nx, nz, ny, periodic_x, 0, d_ph, d_pv, d_cv,
                                      ctx->d_outbox, ctx->d_n_out, ctx->d_n_arr, ctx->d_n_exp,
                                      ctx->d_dest_mask, seed, step, ctx->stream);
    });
    if (rc) return rc;
    // the probabilities live in the staging area, which another call may reallocate: they
    // have been consumed once the stream is idle
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

static inline size_t halo_header_bytes(int row_cells) { return ((size_t)row_cells * 4 + 15) & ~(size_t)15; }

// The strips of one side of the exchange (0: towards the rank below, 1: above, 2: west, 3: east):
// the real boundary strip whose movers travel, the ghost strip behind it they are addressed to, and
// the bit by which a cell of the ghost strip says "I send to the real strip" (k_trans_decide's code).
struct HaloSide { Strip real, ghost; uint64_t bit_in; };
static HaloSide halo_side(const pmcx_ctx *ctx, int side)
{
    const int nx = ctx->halo_nx, nz = ctx->halo_nz, ny = ctx->halo_ny, row = nx * nz, col = ny * nz;
    switch (side) {
    case 0: return {{1 * row, 1, row}, {0, 1, row}, 4ull};                       // ghost row 0 sends north
    case 1: return {{(ny - 2) * row, 1, row}, {(ny - 1) * row, 1, row}, 8ull};   // ghost row ny-1 sends south
    case 2: return {{1, nx, col}, {0, nx, col}, 1ull};                           // ghost column 0 sends east
    default: return {{nx - 2, nx, col}, {nx - 1, nx, col}, 2ull};                // ghost column nx-1 sends west
    }
}
static inline int halo_sides(const pmcx_ctx *ctx) { return ctx->st.trans_cut_x ? 4 : 2; }

int pmcx_transport_export_size(pmcx_ctx *ctx, int32_t side, int64_t *n_bytes)
{
    if (!ctx || !n_bytes || !ctx->halo_nx || side < 0 || side >= halo_sides(ctx))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_export_size: bad argument or no pmcx_transport_begin");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const HaloSide h = halo_side(ctx, side);
    const int L = ctx->halo_row_cells, n = h.real.n;
    int32_t *counts = ctx->d_halo + (size_t)side * 2 * (L + 1), *offsets = counts + L;
    HIP_TRY(ctx, launch_halo_count(ctx->st, ctx->halo_nx, ctx->halo_nz, ctx->halo_ny, ctx->d_outbox, ctx->d_n_out,
                                   h.real, h.ghost, counts, offsets, ctx->stream));
    int32_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, offsets + n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->halo_total[side] = total;
    *n_bytes = (int64_t)halo_header_bytes(n) + (int64_t)total * (int64_t)halo_record_bytes(ctx->st);
    return PMCX_OK;
}

int pmcx_transport_export(pmcx_ctx *ctx, int32_t side, void *dev_buf)
{
    if (!ctx || !dev_buf || !ctx->halo_nx || side < 0 || side >= halo_sides(ctx) || ctx->halo_total[side] < 0)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_export: call pmcx_transport_export_size first");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const HaloSide h = halo_side(ctx, side);
    const int L = ctx->halo_row_cells, n = h.real.n;
    const int32_t *counts = ctx->d_halo + (size_t)side * 2 * (L + 1), *offsets = counts + L;
    HIP_TRY(ctx, hipMemcpyAsync(dev_buf, counts, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, launch_halo_pack(ctx->st, ctx->halo_nx, ctx->halo_nz, ctx->halo_ny, ctx->d_outbox, ctx->d_n_out,
                                  h.real, h.ghost, offsets, (unsigned char *)dev_buf + halo_header_bytes(n),
                                  ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_transport_import(pmcx_ctx *ctx, int32_t side, const void *dev_buf, int64_t n_bytes)
{
    if (!ctx || !ctx->halo_nx || side < 0 || side >= halo_sides(ctx))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: bad argument or no pmcx_transport_begin");
    if (!dev_buf || n_bytes == 0) return PMCX_OK; // no neighbour on that side: the ghost strip stays empty
    if (side >= 2 && ctx->st.trans_gnx == 2)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: a periodic direction two cells wide cannot be cut (both "
                                      "neighbours of a cell are one cell, whose movers the whole domain takes as one "
                                      "sequence): use pmcx_transport on the whole domain");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const HaloSide h = halo_side(ctx, side);
    const int L = ctx->halo_row_cells, row = h.ghost.n;
    if (n_bytes < (int64_t)halo_header_bytes(row))
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: buffer shorter than its header");
    int32_t *offsets = ctx->d_halo + 8 * (size_t)(L + 1) + (size_t)side * (L + 1);
    // The buffer comes from another process: its counts index this block's slabs.  Checked on the
    // host (counts within a cell's capacity, records within the buffer) before anything is
    // written; destinations, copies and classes are checked record by record in the kernel.
    {
        std::vector<int32_t> h_counts((size_t)row);
        HIP_TRY(ctx, hipMemcpyAsync(h_counts.data(), dev_buf, sizeof(int32_t) * (size_t)row,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int64_t total = 0;
        int32_t most = 0;
        for (int r = 0; r < row; r++) most = h_counts[r] > most ? h_counts[r] : most;
        // (a neighbouring rank whose block has grown may send a cell more than a ghost cell here holds)
        if (most > ctx->cfg.cap && most <= 65280 && ctx->auto_grow)
            if (int grc = pmcx_grow(ctx, most)) return grc;
        for (int r = 0; r < row; r++) {
            if (h_counts[r] < 0 || h_counts[r] > ctx->cfg.cap)
                return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: cell %d of the buffer claims %d "
                                              "particles (capacity %d)", r, h_counts[r], ctx->cfg.cap);
            total += h_counts[r];
        }
        const int64_t need = (int64_t)halo_header_bytes(row) + total * (int64_t)halo_record_bytes(ctx->st);
        if (need > n_bytes)
            return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: %lld records need %lld bytes, the "
                                          "buffer holds %lld", (long long)total, (long long)need,
                        (long long)n_bytes);
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_halo_bad, 0, sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, launch_halo_unpack(ctx->st, ctx->halo_nx, ctx->halo_nz, ctx->halo_ny, ctx->d_outbox, ctx->d_n_out,
                                    ctx->d_n_exp, ctx->d_dest_mask, h.ghost, h.real, h.bit_in,
                                    (const int32_t *)dev_buf, offsets,
                                    (const unsigned char *)dev_buf + halo_header_bytes(row),
                                    ctx->d_halo_bad, ctx->stream));
    int32_t bad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, ctx->d_halo_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bad)
        return fail(ctx, PMCX_EINVAL, "pmcx_transport_import: %d records name a destination, a class "
                                      "or a copy count outside this block (they were dropped)", bad);
    return PMCX_OK;
}

int pmcx_transport_finish(pmcx_ctx *ctx, int32_t *n_moved, int32_t *n_arrived)
{
    if (!ctx || !ctx->halo_nx) return fail(ctx, PMCX_EINVAL, "pmcx_transport_finish: no pmcx_transport_begin");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (int grc = grow_if_needed(ctx, ctx->d_n_exp)) return grc; // (pmcx_set_auto_grow: before anything is gathered)
    {
        ProfScope ps(ctx, "transport");
        HIP_TRY(ctx, launch_transport_finish(ctx->st, ctx->halo_nx, ctx->halo_nz, ctx->halo_ny, ctx->halo_px,
                                             0, ctx->d_halo_cv, ctx->d_outbox, ctx->d_n_out, ctx->d_n_arr,
                                             ctx->d_n_exp, ctx->d_dest_mask, 1, ctx->halo_ny - 2,
                                             ctx->stream));
    }
    if (n_moved) HIP_TRY(ctx, hipMemcpyAsync(n_moved, ctx->d_n_out, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    if (n_arrived) HIP_TRY(ctx, hipMemcpyAsync(n_arrived, ctx->d_n_arr, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells, false)); // (the kernels keep the sort keys current)
    ctx->halo_nx = 0;
    return PMCX_OK;
}

int pmcx_condense(pmcx_ctx *ctx, double del_t, int32_t n_sub)
{
    if (!ctx) return PMCX_EINVAL;
    if (ctx->cfg.i_water < 0) return fail(ctx, PMCX_ESTATE, "pmcx_condense: config has no water species (i_water < 0)");
    if (!ctx->kappa_set || !ctx->env_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_condense: kappa (pmcx_set_aero_data) and env (pmcx_set_env) must be set");
    if (!(del_t >= 0) || n_sub < 1) return fail(ctx, PMCX_EINVAL, "pmcx_condense: bad del_t or n_sub");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ProfScope ps(ctx, "condense");
    HIP_TRY(ctx, launch_condense(ctx->st, ctx->d_kappa, ctx->d_rh, del_t, n_sub, ctx->stream));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells));
    return PMCX_OK;
}

int pmcx_diagnostics(pmcx_ctx *ctx, double *tot_num_conc, double *tot_mass_conc, double *num_dist,
                     double *mass_dist)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells, nd = (size_t)ctx->cfg.n_dbin;
    double *d_tn = ctx->d_diag, *d_tm = d_tn + nc, *d_nd = d_tm + nc, *d_md = d_nd + nc * nd;
    {
        ProfScope ps(ctx, "diagnostics");
        HIP_TRY(ctx, launch_diagnostics(ctx->st, d_tn, d_tm, d_nd, d_md, ctx->stream));
    }
    if (tot_num_conc) HIP_TRY(ctx, hipMemcpyAsync(tot_num_conc, d_tn, sizeof(double) * nc, hipMemcpyDeviceToHost, ctx->stream));
    if (tot_mass_conc) HIP_TRY(ctx, hipMemcpyAsync(tot_mass_conc, d_tm, sizeof(double) * nc, hipMemcpyDeviceToHost, ctx->stream));
    if (num_dist) HIP_TRY(ctx, hipMemcpyAsync(num_dist, d_nd, sizeof(double) * nc * nd, hipMemcpyDeviceToHost, ctx->stream));
    if (mass_dist) HIP_TRY(ctx, hipMemcpyAsync(mass_dist, d_md, sizeof(double) * nc * nd, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_diagnostics_ext(pmcx_ctx *ctx, double *species_mass_conc, double *num_conc_by_source,
                         double *pm_mass_conc, int32_t *n_parts)
{
    if (!ctx) return PMCX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells, S = (size_t)ctx->cfg.n_spec,
                 NS = (size_t)ctx->cfg.n_source;
    int rc = ensure_stage(ctx, sizeof(double) * nc * (S + NS + 3) + 64);
    if (rc) return rc;
    double *d_sm = (double *)ctx->stage, *d_sn = d_sm + nc * S, *d_pm = d_sn + nc * NS;
    {
        ProfScope ps(ctx, "diagnostics");
        HIP_TRY(ctx, launch_diagnostics_ext(ctx->st, d_sm, d_sn, d_pm, ctx->stream));
    }
    if (species_mass_conc) HIP_TRY(ctx, hipMemcpyAsync(species_mass_conc, d_sm, sizeof(double) * nc * S, hipMemcpyDeviceToHost, ctx->stream));
    if (num_conc_by_source && NS) HIP_TRY(ctx, hipMemcpyAsync(num_conc_by_source, d_sn, sizeof(double) * nc * NS, hipMemcpyDeviceToHost, ctx->stream));
    if (pm_mass_conc) HIP_TRY(ctx, hipMemcpyAsync(pm_mass_conc, d_pm, sizeof(double) * nc * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (n_parts) HIP_TRY(ctx, hipMemcpyAsync(n_parts, ctx->st.n_part, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_generate_lognormal(pmcx_ctx *ctx, int32_t n_part, int32_t n_mode, const double *mode_frac,
                            const double *gm_radius, const double *gsd_log10,
                            const double *vol_frac, const int32_t *mode_class,
                            const int32_t *mode_source, uint64_t seed, double create_time)
{
    if (!ctx || !mode_frac || !gm_radius || !gsd_log10 || !vol_frac)
        return fail(ctx, PMCX_EINVAL, "pmcx_generate_lognormal: null argument");
    if (!ctx->aero_data_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_aero_data must come first");
    if (n_part < 0 || n_part > ctx->cfg.cap || n_mode < 1 || n_mode > 64)
        return fail(ctx, PMCX_EINVAL, "n_part or n_mode out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const int S = ctx->cfg.n_spec;
    std::vector<double> h(3 * (size_t)n_mode + (size_t)n_mode * S);
    double acc = 0, tot = 0;
    for (int m = 0; m < n_mode; m++) tot += mode_frac[m];
    for (int m = 0; m < n_mode; m++) {
        acc += mode_frac[m];
        h[m] = acc / tot;
        h[n_mode + m] = det_log(gm_radius[m]);
        h[2 * n_mode + m] = gsd_log10[m] * LN_10;
        for (int s = 0; s < S; s++) h[3 * n_mode + (size_t)m * S + s] = vol_frac[(size_t)m * S + s];
    }
    std::vector<int32_t> hi(2 * (size_t)n_mode);
    for (int m = 0; m < n_mode; m++) {
        hi[m] = mode_class ? mode_class[m] : 0;
        hi[n_mode + m] = mode_source ? mode_source[m] : 0;
        if (hi[m] < 0 || hi[m] >= ctx->cfg.n_class)
            return fail(ctx, PMCX_EINVAL, "mode_class[%d] out of range", m);
    }
    int rc = ensure_stage(ctx, sizeof(double) * h.size() + sizeof(int32_t) * hi.size() + 64);
    if (rc) return rc;
    double *dd = (double *)ctx->stage;
    int32_t *di = (int32_t *)(dd + h.size());
    HIP_TRY(ctx, hipMemcpyAsync(dd, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(di, hi.data(), sizeof(int32_t) * hi.size(), hipMemcpyHostToDevice, ctx->stream));
    GenSpecDev g;
    g.n_mode = n_mode;
    g.mode_frac = dd; g.ln_gm_radius = dd + n_mode; g.ln_gsd = dd + 2 * n_mode;
    g.vol_frac = dd + 3 * n_mode;
    g.mode_class = di; g.mode_source = di + n_mode;
    {
        ProfScope ps(ctx, "generate");
        HIP_TRY(ctx, launch_generate(ctx->st, g, n_part, seed, create_time, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, invalidate_sort(ctx, 0, ctx->cfg.n_cells));
    return PMCX_OK;
}

int pmcx_set_cell_id_base(pmcx_ctx *ctx, uint32_t cell_id_base)
{
    if (!ctx) return PMCX_EINVAL;
    ctx->cfg.cell_id_base = cell_id_base;
    ctx->st.cell_id_base = cell_id_base; // (the state view is passed by value with every launch)
    return PMCX_OK;
}

int pmcx_set_gas_state(pmcx_ctx *ctx, int32_t n_gas, const double *mix_rat)
{
    if (!ctx || !mix_rat || n_gas < 1) return fail(ctx, PMCX_EINVAL, "pmcx_set_gas_state: bad argument");
    if (ctx->n_gas && ctx->n_gas != n_gas)
        return fail(ctx, PMCX_EINVAL, "pmcx_set_gas_state: n_gas was %d, now %d", ctx->n_gas, n_gas);
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (!ctx->d_gas) {
        int rc = dev_alloc(ctx, &ctx->d_gas, nc * (size_t)n_gas, false);
        if (!rc) rc = dev_alloc(ctx, &ctx->d_height, 2 * nc, false);
        if (rc) return rc;
        ctx->n_gas = n_gas;
        ctx->gas_scn.n_gas = n_gas;
    }
    return copy_cells_h2d(ctx, ctx->d_gas, mix_rat, sizeof(double) * nc * (size_t)n_gas);
}

int pmcx_get_gas_state(pmcx_ctx *ctx, double *mix_rat)
{
    if (!ctx || !mix_rat) return fail(ctx, PMCX_EINVAL, "pmcx_get_gas_state: null argument");
    if (!ctx->d_gas) return fail(ctx, PMCX_ESTATE, "pmcx_get_gas_state: no gas state was set");
    return copy_cells_d2h(ctx, mix_rat, ctx->d_gas,
                          sizeof(double) * (size_t)ctx->cfg.n_cells * (size_t)ctx->n_gas);
}

int pmcx_set_gas_scenario(pmcx_ctx *ctx, int32_t n_time_e, const double *emission_time,
                          const double *emission_rate_scale, const double *emission,
                          int32_t n_time_d, const double *dilution_time,
                          const double *dilution_rate, const double *background)
{
    if (!ctx) return PMCX_EINVAL;
    if (!ctx->d_gas) return fail(ctx, PMCX_ESTATE, "pmcx_set_gas_scenario: pmcx_set_gas_state must come first");
    if (n_time_e < 0 || n_time_d < 0 || n_time_e > 4096 || n_time_d > 4096 ||
        (n_time_e > 0 && (!emission_time || !emission_rate_scale || !emission)) ||
        (n_time_d > 0 && (!dilution_time || !dilution_rate || !background)))
        return fail(ctx, PMCX_EINVAL, "pmcx_set_gas_scenario: bad series length or null array");
    for (int k = 1; k < n_time_e; k++)
        if (!(emission_time[k] > emission_time[k - 1]))
            return fail(ctx, PMCX_EINVAL, "pmcx_set_gas_scenario: emission_time must increase");
    for (int k = 1; k < n_time_d; k++)
        if (!(dilution_time[k] > dilution_time[k - 1]))
            return fail(ctx, PMCX_EINVAL, "pmcx_set_gas_scenario: dilution_time must increase");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (void *p : ctx->gas_scn_allocs) (void)hipFree(p);
    ctx->gas_scn_allocs.clear();
    GasScenarioDev g{};
    g.n_gas = ctx->n_gas;
    const size_t nc = (size_t)ctx->cfg.n_cells, ng = (size_t)ctx->n_gas;
    auto up = [&](const double *h, size_t count, const double *&d) -> int {
        double *q = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&q, sizeof(double) * count));
        ctx->gas_scn_allocs.push_back(q);
        HIP_TRY(ctx, hipMemcpyAsync(q, h, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
        d = q;
        return PMCX_OK;
    };
    int rc = PMCX_OK;
    if (n_time_e > 0) {
        if (!rc) rc = up(emission_time, (size_t)n_time_e, g.emission_time);
        if (!rc) rc = up(emission_rate_scale, (size_t)n_time_e, g.emission_rate_scale);
        if (!rc) rc = up(emission, nc * (size_t)n_time_e * ng, g.emission);
        g.n_time_e = n_time_e;
    }
    if (n_time_d > 0) {
        if (!rc) rc = up(dilution_time, (size_t)n_time_d, g.dilution_time);
        if (!rc) rc = up(dilution_rate, nc * (size_t)n_time_d, g.dilution_rate);
        if (!rc) rc = up(background, nc * (size_t)n_time_d * ng, g.background);
        g.n_time_d = n_time_d;
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gas_scn = g;
    return PMCX_OK;
}

int pmcx_scenario_update_gas_state(pmcx_ctx *ctx, double dt, double elapsed_time,
                                   const double *height, const double *old_height)
{
    if (!ctx || !height) return fail(ctx, PMCX_EINVAL, "pmcx_scenario_update_gas_state: null argument");
    if (!ctx->d_gas) return fail(ctx, PMCX_ESTATE, "pmcx_scenario_update_gas_state: no gas state was set");
    if (!ctx->env_set) return fail(ctx, PMCX_ESTATE, "pmcx_set_env must come first");
    if (!(dt >= 0)) return fail(ctx, PMCX_EINVAL, "dt must be non-negative");
    const size_t nc = (size_t)ctx->cfg.n_cells;
    for (size_t k = 0; k < nc; k++)
        if (!(height[k] > 0) || (old_height && !(old_height[k] > 0)))
            return fail(ctx, PMCX_EINVAL, "pmcx_scenario_update_gas_state: heights must be positive (cell %zu)", k);
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    // (pageable host memory: the copies have left the caller's arrays when these calls return)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_height, height, sizeof(double) * nc, hipMemcpyHostToDevice, ctx->stream));
    if (old_height)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_height + nc, old_height, sizeof(double) * nc,
                                    hipMemcpyHostToDevice, ctx->stream));
    ProfScope ps(ctx, "gas_scenario");
    HIP_TRY(ctx, launch_gas_update(ctx->gas_scn, ctx->d_gas, ctx->d_temp, ctx->d_pressure, ctx->d_height,
                                   old_height ? ctx->d_height + nc : nullptr, ctx->cfg.n_cells, dt,
                                   elapsed_time, ctx->stream, ctx->st.loop_mask));
    return PMCX_OK;
}

void pmcx_step_flags_default(pmcx_step_flags *f)
{
    memset(f, 0, sizeof *f);
    f->struct_size = (int32_t)sizeof(pmcx_step_flags);
    f->do_coagulation = 1; f->allow_doubling = 1; f->allow_halving = 1;
    f->kernel_type = PMCX_COAG_KERNEL_TYPE_BROWN;
    f->partmc_chem_dt = 60.0;
}

int pmcx_set_loop_cells(pmcx_ctx *ctx, const int32_t *in_loop)
{
    if (!ctx) return PMCX_EINVAL;
    if (!in_loop) { ctx->st.loop_mask = nullptr; return PMCX_OK; }
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nc = (size_t)ctx->cfg.n_cells;
    if (!ctx->d_loop_mask) {
        int rc = dev_alloc(ctx, &ctx->d_loop_mask, nc);
        if (rc) return rc;
    }
    std::vector<uint8_t> h(nc);
    for (size_t k = 0; k < nc; k++) h[k] = in_loop[k] ? 1 : 0;
    int rc = copy_cells_h2d(ctx, ctx->d_loop_mask, h.data(), nc);
    if (rc) return rc;
    ctx->st.loop_mask = ctx->d_loop_mask;
    return PMCX_OK;
}

int32_t pmcx_max_cap(const pmcx_config *cfg)
{
    if (!cfg || cfg->n_class < 1 || cfg->n_bin < 1 || (int64_t)cfg->n_class * cfg->n_bin > 65535) return 0;
    // slots are 16-bit list entries (0xFFFF marks none), cap a multiple of 64: whatever the number of
    // weight classes -- the lists' layout adapts to the room perm has (pmcx.h, pmcx_get_sort)
    return 65280;
}

int pmcx_timestep(pmcx_ctx *ctx, const pmcx_step_flags *f)
{
    if (!ctx || !f) return fail(ctx, PMCX_EINVAL, "pmcx_timestep: null argument");
    if (f->struct_size != (int32_t)sizeof(pmcx_step_flags))
        return fail(ctx, PMCX_EINVAL, "pmcx_step_flags size mismatch (%d vs %zu): built against another "
                                      "include/pmcx.h? (pmcx_step_flags_default sets it)",
                    f->struct_size, sizeof(pmcx_step_flags));
    if (!(f->dt > 0) || !(f->partmc_chem_dt > 0) || !(f->elapsed_time >= 0))
        return fail(ctx, PMCX_EINVAL, "pmcx_timestep: dt and partmc_chem_dt must be positive");
    if (!ctx->n_ideal_set)
        return fail(ctx, PMCX_ESTATE, "pmcx_timestep: pmcx_set_n_part_ideal must come first");
    if (f->sea_salt_opt > 0 && (!f->height || !f->ocean_surface || !f->wind_speed))
        return fail(ctx, PMCX_EINVAL, "pmcx_timestep: sea salt needs height, ocean_surface and wind_speed");
    // chemistry-step gate, interface/wrf_pmc_driver.F90:189
    const bool chem_step = std::fmod(f->elapsed_time, f->partmc_chem_dt) == 0.0;
    const uint32_t step = (uint32_t)(f->elapsed_time / f->partmc_chem_dt);
    const uint32_t wrf_step = (uint32_t)(f->elapsed_time / f->dt);
    int rc = PMCX_OK;
    // one status word for the step: every stage adds its bits (each of them, called by itself,
    // would start a new one and wipe what mc_coag reported)
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    // pmcx_set_auto_grow: room for this step's emissions, background and sea salt (the fullest cell
    // beyond its fill mark: the block grows before the step, as PartMC's particle arrays grow on use)
    rc = grow_if_needed(ctx, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * (size_t)ctx->cfg.n_cells, ctx->stream));
    struct Hold { pmcx_ctx *c; Hold(pmcx_ctx *x) : c(x) { c->hold_status = true; } ~Hold() { c->hold_status = false; } } hold(ctx);
    if (chem_step && f->do_coagulation) {                                        // driver:189-198
        rc = pmcx_mc_coag(ctx, f->kernel_type, f->partmc_chem_dt, f->random_seed, step);
        if (rc) return rc;
    }
    if (f->height && ctx->d_gas) {                                               // driver:202-205
        rc = pmcx_scenario_update_gas_state(ctx, f->dt, f->elapsed_time, f->height, f->old_height);
        if (rc) return rc;
    }
    if (f->aero_emission || f->aero_dilution_rate || f->aero_background) {       // driver:207-210
        rc = pmcx_scenario_update_aero_state(ctx, f->dt, f->elapsed_time, f->random_seed, wrf_step,
                                             f->aero_emission, f->aero_dilution_rate,
                                             f->aero_background, nullptr, nullptr, nullptr, f->height, f->old_height);
        if (rc) return rc;
    }
    if (f->sea_salt_opt > 0) {                                                   // driver:212-223
        rc = pmcx_seasalt_emissions(ctx, f->dt, f->elapsed_time, f->random_seed, wrf_step, f->sea_salt_opt,
                                    f->i_na, f->i_cl, f->ocean_surface, f->wind_speed, f->height,
                                    f->ss_mode_class, f->ss_mode_source, nullptr, nullptr);
        if (rc) return rc;
    }
    return pmcx_rebalance(ctx, f->allow_doubling, f->allow_halving, f->random_seed, wrf_step); // :248-250
}

int pmcx_get_bin_tables(pmcx_ctx *ctx, double *v_edge, double *v_samp)
{
    if (!ctx) return PMCX_EINVAL;
    if (v_edge) memcpy(v_edge, ctx->v_edge.data(), sizeof(double) * ctx->v_edge.size());
    if (v_samp) memcpy(v_samp, ctx->v_samp.data(), sizeof(double) * ctx->v_samp.size());
    return PMCX_OK;
}
int pmcx_get_diag_grid(pmcx_ctx *ctx, double *d_edge, double *d_width)
{
    if (!ctx) return PMCX_EINVAL;
    if (d_edge) memcpy(d_edge, ctx->d_edge.data(), sizeof(double) * ctx->d_edge.size());
    if (d_width) memcpy(d_width, ctx->d_width.data(), sizeof(double) * ctx->d_width.size());
    return PMCX_OK;
}
int pmcx_get_kmax(pmcx_ctx *ctx, int32_t cell, double *kmax)
{
    if (!ctx || !kmax || cell < 0 || cell >= ctx->cfg.n_cells) return fail(ctx, PMCX_EINVAL, "pmcx_get_kmax: bad argument");
    if (!ctx->bounds_kernel) return fail(ctx, PMCX_ESTATE, "kernel bounds not computed yet");
    const size_t bb = (size_t)ctx->cfg.n_bin * ctx->cfg.n_bin;
    // the device table is fp32 (bounds rounded up): widen for the caller
    std::vector<float> f(bb);
    int rc = copy_cells_d2h(ctx, f.data(), ctx->st.kmax + (size_t)cell * ctx->st.kmax_stride,
                            sizeof(float) * bb);
    if (rc) return rc;
    for (size_t k = 0; k < bb; k++) kmax[k] = (double)f[k];
    return PMCX_OK;
}
int pmcx_get_caches(pmcx_ctx *ctx, int32_t cell, double *vtot, double *mass)
{
    if (!ctx || cell < 0 || cell >= ctx->cfg.n_cells) return fail(ctx, PMCX_EINVAL, "pmcx_get_caches: bad argument");
    const size_t cap = (size_t)ctx->cfg.cap;
    int rc = PMCX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, launch_compact(ctx->st, cell, 1, 0, 1, ctx->stream));
    // the two caches sit in the particles' 32-byte records: strided copies
    const Hot *src = ctx->st.hot + (size_t)cell * cap;
    if (vtot) HIP_TRY(ctx, hipMemcpy2DAsync(vtot, sizeof(double), &src->vtot, sizeof(Hot), sizeof(double),
                                            cap, hipMemcpyDeviceToHost, ctx->stream));
    if (mass) HIP_TRY(ctx, hipMemcpy2DAsync(mass, sizeof(double), &src->mass, sizeof(Hot), sizeof(double),
                                            cap, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}
int pmcx_get_sort(pmcx_ctx *ctx, int32_t cell, uint32_t *perm, int32_t *gcnt, int32_t *goff)
{
    if (!ctx || cell < 0 || cell >= ctx->cfg.n_cells) return fail(ctx, PMCX_EINVAL, "pmcx_get_sort: bad argument");
    const size_t pcap = (size_t)ctx->st.pcap, G = (size_t)ctx->cfg.n_class * ctx->cfg.n_bin;
    int rc = PMCX_OK;
    if (perm) { // (the lists hold 16-bit slot numbers; the interface keeps its 32-bit entries)
        std::vector<uint16_t> h(pcap);
        rc = copy_cells_d2h(ctx, h.data(), ctx->st.perm + cell * pcap, sizeof(uint16_t) * pcap);
        for (size_t k = 0; k < pcap; k++) perm[k] = h[k];
    }
    if (!rc && gcnt) rc = copy_cells_d2h(ctx, gcnt, ctx->st.gcnt + cell * G, sizeof(int32_t) * G);
    if (!rc && goff) rc = copy_cells_d2h(ctx, goff, ctx->st.goff + cell * (G + 1), sizeof(int32_t) * (G + 1));
    return rc;
}

int pmcx_test_math(pmcx_ctx *ctx, int32_t op, int64_t n, const double *in, const double *in2,
                   const double *in3, const double *in4, int32_t ktype, double temp,
                   double pressure, double *out)
{
    if (!ctx || !in || !out || n < 1) return fail(ctx, PMCX_EINVAL, "pmcx_test_math: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_stage(ctx, sizeof(double) * 5 * (size_t)n);
    if (rc) return rc;
    double *d = (double *)ctx->stage;
    const double *src[4] = {in, in2, in3, in4};
    for (int k = 0; k < 4; k++)
        if (src[k])
            HIP_TRY(ctx, hipMemcpyAsync(d + k * n, src[k], sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_test_math(op, n, d, d + n, d + 2 * n, d + 3 * n, ktype, temp, pressure,
                                  ctx->cfg.const_k, ctx->cfg.additive_beta, d + 4 * n, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out, d + 4 * n, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_test_philox(pmcx_ctx *ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!ctx || !ctr || !key || !out) return fail(ctx, PMCX_EINVAL, "pmcx_test_philox: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_stage(ctx, 64);
    if (rc) return rc;
    HIP_TRY(ctx, launch_test_philox(ctr, key, (uint32_t *)ctx->stage, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->stage, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PMCX_OK;
}

int pmcx_profile_phase_cycles(pmcx_ctx *ctx, uint64_t out[12], int32_t reset)
{
    if (!ctx || !out) return fail(ctx, PMCX_EINVAL, "pmcx_profile_phase_cycles: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, seq_stamps_read(out, reset != 0));
    return PMCX_OK;
}

int pmcx_profile_enable(pmcx_ctx *ctx, int32_t on)
{
    if (!ctx) return PMCX_EINVAL;
    ctx->prof = on != 0;
    return PMCX_OK;
}
int pmcx_profile_reset(pmcx_ctx *ctx)
{
    if (!ctx) return PMCX_EINVAL;
    (void)hipStreamSynchronize(ctx->stream);
    prof_drain(ctx);
    ctx->profile.clear();
    return PMCX_OK;
}
int pmcx_profile_get(pmcx_ctx *ctx, const char *name, double *ms, int64_t *launches)
{
    if (!ctx || !name) return PMCX_EINVAL;
    prof_drain(ctx);
    auto it = ctx->profile.find(name);
    if (ms) *ms = it == ctx->profile.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->profile.end() ? 0 : it->second.launches;
    return PMCX_OK;
}

} // extern "C"
