// pmcx_kernels.h -- device state view and kernel launchers (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pmcx {

enum : int {
    ST_EVENT_OVERFLOW = 1,
    ST_APPEND_OVERFLOW = 2,
    ST_KMAX_EXCEEDED = 4,
    ST_OUT_OF_GRID = 8,
    ST_CAP_OVERFLOW = 16,
    ST_KMAX_RANGE = 32,
    ST_OUTBOX_OVERFLOW = 64,  // transport, law 1: a source cell's outbox (cap entries) ran out
    ST_TRANS_COPIES = 128     // transport, law 1: an emitter's copy count met the guard (4096)
};

// What the pair sampler reads and writes of a particle -- the cached total volume and mass a
// kernel evaluation needs, the create times a coagulation merges -- is one 32-byte record:
// the sampler is bound by the number of distinct 64-byte lines it touches,
// and a candidate pair now costs one line per particle, events write to lines they have read.
struct __attribute__((aligned(32))) Hot {
    double vtot, mass; // sum_s vol (< 0: tombstone), sum_s density_s vol
    double lct, gct;   // least / greatest create time
};
// What only whole-particle moves touch (and events, when sources are tracked).
struct __attribute__((aligned(16))) Cold {
    int64_t id;
    int32_t n_orig0;   // constituents from source 0 (the other sources: DeviceState::n_orig)
    int32_t pad;
};

#ifndef PMCX_SLIST_CAP
#define PMCX_SLIST_CAP 2048
#endif
constexpr int SLIST_CAP = PMCX_SLIST_CAP;  // entries of a cell's list of possible slots (the last one ends it)
constexpr uint32_t SL_END = 0xFFFFFFFFu, SL_FULL = 0xFFFFFFFEu; // end of the sweep / list full: the next round goes on
constexpr int EVLOG_CAP = 256; // entries of a cell's event log (two per coagulation at most)

// entries of perm per cell: min(65535, cap + cap/8 + min(16 G, 8192)) -- offsets into it are 16 bits in
// the samplers' LDS; the lists' layout adapts to what it holds (sort_slack_level, pmcx_kernels.hip)
inline int32_t sort_pcap(int32_t cap, int32_t G)
{
    const int64_t extra = 16 * (int64_t)G < 8192 ? 16 * (int64_t)G : 8192;
    const int64_t want = (int64_t)cap + cap / 8 + extra;
    return (int32_t)(want < 65535 ? want : 65535);
}

// Plain view of everything a kernel needs; passed by value.
struct DeviceState {
    int32_t n_cells, cap, n_spec, n_class, n_source, n_bin, n_dbin, i_water;
    int32_t max_events, max_appends;
    int32_t compact_den; // holes are closed once more than 1/compact_den of a cell's slots are dead
    int32_t pcap; // entries of perm per cell: sort_pcap(cap, n_class n_bin)
    uint32_t cell_id_base;
    double const_k, additive_beta;
    double rho_min, rho_max;
    float log2_v0, inv_dlog2;  // float estimate of the sort-grid bin from the bits of v
    int32_t bin_fast;          // estimate is within one bin (grid coarse enough)
    float dlog2_d0, dinv_dlog2; // same for the diagnostics diameter grid
    int32_t dbin_fast;
    // model tables
    const double *density; // [n_spec]
    const double *v_edge;  // [n_bin+1]
    const double *v_samp;  // [n_bin*3]
    const double *d_edge;  // [n_dbin+1]
    const double *d_width; // [n_dbin]
    // per cell
    const double *temp, *pressure; // [n_cells]
    double *mag;                   // [n_cells][n_class]
    const double *n_ideal;         // [n_cells][n_class]
    float *kmax;                   // [n_cells][kmax_stride] bounds rounded up to fp32, row b1 = [b1*n_bin ..)
    int32_t kmax_stride;           // n_bin*n_bin rounded up to a multiple of 4 (16-byte loads)
    int32_t *kmax_rng;             // [n_cells][4] {first bin, last bin, kernel type} the bounds of a cell
                                   // were computed for (rows and columns outside are stale)
    double *kmax_env;              // [n_cells][2] temperature and pressure they were computed at
    int32_t *work_list;            // [2 n_cells + 2] scratch: which cells the next kernel has work for
    int32_t *n_part;               // [n_cells] live particles
    int32_t *n_hi;                 // [n_cells] slots in use (live + tombstones)
    int64_t *next_id;              // [n_cells]
    // particles
    double *vol;     // [n_cells][cap][n_spec]: a particle's species are contiguous
    Hot *hot;        // [n_cells][cap] cached total volume and mass, create times
    int32_t *wclass; // [n_cells][cap]
    Cold *cold;      // [n_cells][cap] id, source-0 count
    int32_t *n_orig; // [n_cells][n_source - 1][cap] sources 1.. (source 0 is in meta)
    // scratch / outputs of the coagulation step
    // aero_sorted, kept between calls while sort_valid (aero_state%valid_sort)
    uint16_t *perm;  // [n_cells][pcap] slot lists (slots are below 65 280), group after group, room behind each
    int32_t *gcnt;   // [n_cells][n_class*n_bin] list lengths
    int32_t *goff;   // [n_cells][n_class*n_bin + 1] list starts
    int32_t *sort_valid; // [n_cells]
    // Sort keys, kept between sorts: the group (class * n_bin + bin) of every slot, 0xFFFF for a
    // tombstone.  While key_valid[c], applying the cell's event log (the slots the pair sampler
    // re-binned or removed since the keys were last brought up to date: slot | key << 16) to
    // gkey gives the key of every slot below n_hi -- the sort then streams 2 bytes per particle
    // instead of its record.  Whatever else changes a cell's particles clears key_valid.
    uint16_t *gkey;      // [n_cells][cap]
    int32_t *key_valid;  // [n_cells]
    uint32_t *evlog;     // [n_cells][EVLOG_CAP]
    int32_t *n_log;      // [n_cells]
    int32_t trans_lateral; // pmcx_set_transport_lateral: the reference's open lateral boundaries
    int32_t trans_j0, trans_gny; // a slab's place in the domain: the row its row 0 stands for, the domain's rows (0: the block is the domain)
    // a block cut in x as well (pmcx_set_transport_cuts): columns 0 and nx - 1 are ghosts of the
    // neighbouring ranks; the column its column 0 stands for, the domain's columns
    int32_t trans_cut_x, trans_i0, trans_gnx;
    // the global id of local cell c (the RNG key): cell_id_base + c when a row of the block is a row
    // of the domain (id_nx_local = 0), else cell_id_base + (c / id_nx_local) id_nx_global + c % id_nx_local
    int32_t id_nx_local, id_nx_global;
    // pmcx_set_sampler_variant (diagnostic): 0 = the library's own choice
    int32_t force_lanes, force_wpb, force_stamps;
    uint16_t *seq_ovf;   // [n_cells][2 G + 2 max_appends] lane-per-cell sampler: entries past a group's room
    int32_t *n_events, *n_samp, *n_coag, *status; // [n_cells]
    // pmcx_set_loop_cells: the cells partmc_timestep's loop nest covers (nullptr: every cell)
    const uint8_t *loop_mask;
    // the lane-per-cell sampler with its draw split off (k_coag_lane<.., PRE>): a cell's list of the
    // slots whose count could be non-zero, where an unfinished cell stands, the cells to finish
    uint4 *pre;        // [n_cells][128] {uniform, bound, slot, -} (nullptr: not allocated)
    int32_t *resume;   // [n_cells][4] where a kernel left a cell to the one launched behind it: slot settled last
                       // (-1: none), entries in the overflow table, samples of that slot still to evaluate, the next one's index
    int32_t *work2;    // [1 + n_cells] count, cells
    // several weight classes: the pair sampler's slots (pairs of groups) number n_class^2 n_bin^2 / 2 and
    // nearly all draw a count of zero; k_coag_screen, a wavefront per cell, tests them all side by side
    // and lists the few that might not (k_coag_lane<.., LISTED> then sweeps the list, a lane per cell)
    uint2 *slist;      // [n_cells][SLIST_CAP] {c1 << 24 | b1 << 16 | c2 << 8 | b2, the slot's uniform}, SL_END / SL_FULL behind the last
    uint16_t *gnb;     // [n_cells][n_class n_bin] the length each group was tested with (0: not walked): the list
                       // describes the cell while no group is longer
    int32_t *work3;    // [1 + n_cells] count, cells (the rounds alternate between work2 and work3)
    int32_t *cls_cnt;  // [n_cells][n_class] live particles per class (k_class_counts: the transport's probabilities and preweight)
    int32_t cls_fresh; // 1: cls_cnt holds the populations as they are (the preweight right behind the probabilities)
    // particle%component(:) (interface/wrf_pmc_driver.F90:1421-1437), the first n_comp_max of a
    // particle's list (0: not carried): source ids a byte each (0xFF past the end), create times;
    // the coagulations of the last mc_coag in event order, merged by k_comp_merge behind the sampler
    int32_t n_comp_max;
    // aero_state_prepare_weight_for_add in every add from a distribution (pmcx_set_weight_preparation)
    int32_t trans_law;    // 0: movers carry 1 + geometric copies when r < 1; 1: movers arrive once, emitters send the rest
    int32_t prep_on, prep_doubling, prep_halving;
    int32_t *prep_resume; // [1 + 5 n_cells]: count, then {cell, position, n_emit, n_dil_in, n_dil_out} (k_scenario_update)
    uint64_t *comp_src;  // [n_cells][cap]
    double *comp_time;   // [n_cells][cap][n_comp_max]
    uint64_t *mlog;      // [n_cells][max_events] pn | po << 16 | (pn is the first candidate) << 32
};

struct GenSpecDev {
    int32_t n_mode;
    const double *mode_frac, *ln_gm_radius, *ln_gsd, *vol_frac;
    const int32_t *mode_class, *mode_source;
    int32_t per_cell; // 1: ln_gm_radius .. mode_source are [n_cells][n_mode] (k_scenario_update, k_boundary_resample)
};

// the gas half of scenario_t, resident (pmcx_set_gas_scenario)
struct GasScenarioDev {
    int32_t n_gas, n_time_e, n_time_d;
    const double *emission_time, *emission_rate_scale; // [n_time_e]
    const double *emission;                            // [n_cells][n_time_e][n_gas]
    const double *dilution_time;                       // [n_time_d]
    const double *dilution_rate;                       // [n_cells][n_time_d]
    const double *background;                          // [n_cells][n_time_d][n_gas]
};

// shared-memory footprints (host side needs them to size launches)
size_t bin_sort_lds_bytes(const DeviceState &st);
size_t sampler_lds_bytes_per_cell(const DeviceState &st);
size_t diagnostics_lds_bytes(const DeviceState &st);
size_t compact_lds_bytes(const DeviceState &st);

hipError_t launch_scatter_f64(const double *src, double *dst, const int64_t *part_offset, int n_up,
                              int cap, int64_t dst_cell_stride, int64_t src_row_stride, int n_rows,
                              int64_t dst_row_stride, hipStream_t s, int elem_stride = 1);
hipError_t launch_scatter_i32(const int32_t *src, int32_t *dst, const int64_t *part_offset,
                              int n_up, int cap, int64_t dst_cell_stride, int64_t src_row_stride,
                              int n_rows, int64_t dst_row_stride, hipStream_t s, int elem_stride = 1);
hipError_t launch_gather_f64(double *dst, const double *src, const int64_t *part_offset, int n_dn,
                             int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s, int elem_stride = 1);
hipError_t launch_gather_i32(int32_t *dst, const int32_t *src, const int64_t *part_offset,
                             int n_dn, int64_t src_cell_stride, int64_t dst_row_stride, int n_rows,
                             int64_t src_row_stride, hipStream_t s, int elem_stride = 1);
hipError_t launch_set_n_part(int32_t *n_part, int32_t *n_hi, const int64_t *part_offset, int n_up,
                             hipStream_t s);
hipError_t launch_compact(const DeviceState &st, int cell_first, int n_cells_do, int num, int den,
                          hipStream_t s);
hipError_t launch_refresh_caches(const DeviceState &st, int cell_first, int n_cells_do,
                                 hipStream_t s);
hipError_t launch_kernel_bounds(const DeviceState &st, int ktype, int margin, int32_t *scratch,
                                hipStream_t s);
hipError_t launch_bin_sort(const DeviceState &st, hipStream_t s);
hipError_t launch_coag_sample(const DeviceState &st, int ktype, double del_t, uint64_t seed,
                              uint32_t step, hipStream_t s);
bool coag_sample_splits_draw(const DeviceState &st, int ktype); // the form launch_coag_sample picks
int coag_sample_form(const DeviceState &st, int ktype);  // the form's number (pmcx_set_sampler_variant's lanes)
int coag_listed_rounds();                                // form 5: rounds launched by the last call (the last one empty)
hipError_t seq_stamps_read(uint64_t out[12], bool reset); // diagnostic (pmcx_set_sampler_variant, stamps = 1)
hipError_t launch_comp_upload(const DeviceState &st, int cell_first, int n_up, const int64_t *part_offset,
                              const int32_t *len, const int32_t *source, const double *time, hipStream_t s);
hipError_t launch_comp_download(const DeviceState &st, int cell_first, int n_dn, const int64_t *part_offset,
                                int32_t *len, int32_t *source, double *time, hipStream_t s);
hipError_t launch_comp_default(const DeviceState &st, int cell_first, int n_up, hipStream_t s);
hipError_t launch_rebalance(const DeviceState &st, int allow_doubling, int allow_halving,
                            uint64_t seed, uint32_t step, hipStream_t s);
hipError_t launch_diagnostics(const DeviceState &st, double *tot_num, double *tot_mass,
                              double *num_dist, double *mass_dist, hipStream_t s);
hipError_t launch_diagnostics_ext(const DeviceState &st, double *species_mass, double *source_num,
                                  double *pm, hipStream_t s);
hipError_t launch_generate(const DeviceState &st, const GenSpecDev &g, int n, uint64_t seed,
                           double create_time, hipStream_t s);
hipError_t launch_scenario_update(const DeviceState &st, double dt, double create_time, uint64_t seed,
                                  uint32_t step, const GenSpecDev *em, const double *em_conc,
                                  const double *dilution_rate, const GenSpecDev *bg,
                                  const double *bg_conc, int32_t *n_emit, int32_t *n_dil_in,
                                  int32_t *n_dil_out, hipStream_t s, uint32_t emit_kind = 0,
                                  const double *height = nullptr, const double *old_height = nullptr);
// (emit_kind: the RNG stream of the emission draws -- 0 the scenario's emission, 1 is the
// background's, 2 the sea-salt source -- so that sources acting in one step stay independent)
hipError_t launch_gas_update(const GasScenarioDev &g, double *mix_rat, const double *temp,
                             const double *pressure, const double *height, const double *old_height,
                             int n_cells, double dt, double elapsed_time, hipStream_t s,
                             const uint8_t *loop_mask = nullptr);
size_t transport_lds_bytes(const DeviceState &st, int nz);
hipError_t launch_transport(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                            int periodic_y, const double *prob_h, const double *prob_v,
                            const double *cell_volume, void *outbox, int32_t *n_out, int32_t *n_arr,
                            int32_t *n_exp, uint64_t *dest_mask, uint64_t seed, uint32_t step,
                            hipStream_t s);
// a strip of cells: cell(r) = base + r * stride, r < n (a row of the block: stride 1; a column: stride nx)
struct Strip { int32_t base, stride, n; };
// counts[n] and offsets[n + 1] of the movers of strip src bound for strip dst; their records; the
// receiving side (ghost strip <- the neighbour's boundary strip, destinations in strip real;
// mask_bit: the direction ghost -> real as k_trans_decide codes it)
hipError_t launch_halo_count(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                             Strip src, Strip dst, int32_t *counts, int32_t *offsets, hipStream_t s);
hipError_t launch_halo_pack(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                            Strip src, Strip dst, const int32_t *offsets, void *recs, hipStream_t s);
hipError_t launch_halo_unpack(const DeviceState &st, int nx, int nz, int ny, void *outbox, int32_t *n_out,
                              int32_t *n_exp, uint64_t *dest_mask, Strip ghost, Strip real, uint64_t mask_bit,
                              const int32_t *counts, int32_t *offsets, const void *recs, int32_t *bad,
                              hipStream_t s);
hipError_t launch_transport_begin(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                                  int periodic_y, const double *prob_h, const double *prob_v,
                                  const double *cell_volume, void *outbox, int32_t *n_out,
                                  int32_t *n_arr, int32_t *n_exp, uint64_t *dest_mask, uint64_t seed,
                                  uint32_t step, hipStream_t s);
hipError_t launch_transport_finish(const DeviceState &st, int nx, int nz, int ny, int periodic_x,
                                   int periodic_y, const double *cell_volume, void *outbox,
                                   int32_t *n_out, int32_t *n_arr, int32_t *n_exp,
                                   uint64_t *dest_mask, int row_first, int n_rows, hipStream_t s);
// transport probabilities (k_trans_probs); every pointer is device memory
struct TransFieldsArgs {
    int32_t nx, nz, ny, do_vertmix, all_columns, i_offset, j_offset, global_nx, global_ny;
    double dt, dx, dy;
    const float *alt, *xkhh, *u_flux, *v_flux, *w_flux, *z, *z_at_w, *exch_h;
    double *prob_h, *prob_v;
    int32_t *n_substeps, *bad, *n_class_part;
    int32_t *clamp_flag;
};
hipError_t launch_boundary_copy(const DeviceState &st, int n_pairs, const int32_t *from, const int32_t *to,
                                hipStream_t s);
hipError_t launch_boundary_resample(const DeviceState &st, const int32_t *action, const GenSpecDev &bg,
                                    const double *bg_conc, const double *density, double create_time,
                                    uint64_t seed, uint32_t step, int32_t *n_added, int32_t *n_removed,
                                    hipStream_t s);
hipError_t launch_preweight_slab_begin(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                       int j_global0, int global_ny, const double *prob_h,
                                       const double *prob_v, const double *cell_volume, double *old,
                                       double *edge_out, hipStream_t s);
hipError_t launch_preweight_slab_finish(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                        int j_global0, int global_ny, int lo_edge, int hi_edge,
                                        const double *prob_h, const double *prob_v,
                                        const double *cell_volume, double *old, double *expect,
                                        const double *gin_lo, const double *gin_hi, int32_t *n_removed,
                                        uint64_t seed, uint32_t step, hipStream_t s, int w_edge = 0,
                                        int e_edge = 0, const double *gin_w = nullptr, const double *gin_e = nullptr);
hipError_t launch_transport_preweight(const DeviceState &st, int nx, int nz, int ny, int ideal,
                                      const double *prob_h, const double *prob_v,
                                      const double *cell_volume, double *old, double *expect,
                                      int32_t *n_removed, uint64_t seed, uint32_t step, hipStream_t s);
size_t trans_probs_lds_bytes(const DeviceState &st, int nz);
hipError_t launch_transport_probs(const DeviceState &st, const TransFieldsArgs &a, hipStream_t s);
size_t halo_record_bytes(const DeviceState &st);
// per_cell6: [6][n_cells] aer_res_a, ustar, alpha, big_a, height, inv_density
hipError_t launch_dry_deposition(const DeviceState &st, double dt, uint64_t seed, uint32_t step,
                                 const int32_t *surface, const double *per_cell6,
                                 int32_t *n_removed, int32_t *unrealistic, hipStream_t s);
hipError_t launch_condense(const DeviceState &st, const double *kappa, const double *rel_humid,
                           double del_t, int n_sub, hipStream_t s);
hipError_t launch_totals(const DeviceState &st, int64_t *out4, hipStream_t s);
// out3[0] = max over cells of n_part + n_exp (n_exp may be null), out3[1] = max n_hi, out3[2] = the
// cells' status words or-ed
hipError_t launch_need_max(const DeviceState &st, const int32_t *n_exp, int32_t *out3, hipStream_t s);
// a k_trans_decide whose outboxes ran out (ST_OUTBOX_OVERFLOW) taken back: the movers' sort keys restored,
// the bit cleared -- so that the decision can be taken again with more room
hipError_t launch_transport_undo(const DeviceState &st, void *outbox, const int32_t *n_out, hipStream_t s);
// out[0] = max over the cells of an upper bound of the slots a scenario update (launch_scenario_update
// with the same arguments) can bring a cell to: its slots in use, its classes' ideal counts (what the
// weight preparation of an add can copy), and per mode the Poisson mean of the sample + 6 sigma + 12
hipError_t launch_adds_bound(const DeviceState &st, double dt, double elapsed_time, const GenSpecDev *em,
                             const double *em_conc, const double *dilution_rate, const GenSpecDev *bg,
                             const double *bg_conc, const double *height, const double *old_height, int32_t *out,
                             hipStream_t s);
// the slots [old_cap, new_cap) of every cell of a block that has just grown: dead
hipError_t launch_grow_tail(const DeviceState &st, int old_cap, hipStream_t s);
hipError_t launch_test_math(int op, int64_t n, const double *a, const double *b, const double *c3,
                            const double *c4, int ktype, double temp, double pressure,
                            double const_k, double additive_beta, double *out, hipStream_t s);
hipError_t launch_test_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t *out_dev,
                              hipStream_t s);

} // namespace pmcx
