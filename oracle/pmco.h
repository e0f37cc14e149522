/*
 * pmco.h -- ORACLE (test infrastructure, not product code).
 *
 * Plain-C, serial, CPU restatement of the per-grid-cell microphysics path
 * that WRF-PartMC runs in partmc_timestep
 * (/root/reference/interface/wrf_pmc_driver.F90:46-290):
 *
 *     mc_coag(COAG_KERNEL_TYPE_BROWN, env_state, aero_data, aero_state,
 *             del_t, n_samp, n_coag)                 wrf_pmc_driver.F90:192-194
 *     aero_state_rebalance(...)                      wrf_pmc_driver.F90:248-250
 *     partmc_process observables                     wrf_pmc_driver.F90:783,838-857
 *
 * PARITY UNPINNED: the arithmetic of mc_coag lives in the git submodule
 * compdyn/partmc (.gitmodules:1-3), which is not vendored in /root/reference
 * (partmc/ is empty, pinned commit unknown), and the reference holds no test,
 * golden vector or fixture for this path.  This file
 * therefore restates the *published* PartMC algorithm (Riemer et al. 2009,
 * J. Geophys. Res. 114 D09202; DeVille et al. 2011, J. Comput. Phys. 230
 * 8427; Jacobson 2005 eq. 15.33 for the Brownian kernel) behind the call
 * signature and data fields the interface uses, and is pinned by analytic
 * Smoluchowski solutions, conservation laws, Random123 known-answer vectors
 * and an independent transcription of the in-tree Fuchs-form kernel
 * (WRFV3/chem/module_mosaic_coag.F:746-757,898-936); see tests/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include, link or call anything under oracle/.
 */
#ifndef PMCO_H
#define PMCO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Coagulation kernel ids.  1..7 follow PartMC's COAG_KERNEL_TYPE_* numbering
 * (the interface selects COAG_KERNEL_TYPE_BROWN, wrf_pmc_driver.F90:192; the
 * objects of the others are linked, interface/Makefile:24-30). */
enum {
    PMCO_KERNEL_SEDI = 1,
    PMCO_KERNEL_ADDITIVE = 2,
    PMCO_KERNEL_CONSTANT = 3,
    PMCO_KERNEL_BROWN = 4,
    PMCO_KERNEL_ZERO = 5,
    PMCO_KERNEL_BROWN_SEDI = 101 /* Brownian + sedimentation (BASELINE config 2) */
};

/* cell status bits */
enum {
    PMCO_OK = 0,
    PMCO_ST_EVENT_OVERFLOW = 1,  /* event log full: step stopped early        */
    PMCO_ST_APPEND_OVERFLOW = 2, /* list-append log full: step stopped early  */
    PMCO_ST_KMAX_EXCEEDED = 4,   /* some K/K_max > 1 (bound too tight)        */
    PMCO_ST_OUT_OF_GRID = 8,     /* a particle volume fell outside the bins   */
    PMCO_ST_CAP_OVERFLOW = 16,   /* rebalance doubling exceeded capacity      */
    PMCO_ST_OUTBOX_OVERFLOW = 64, /* transport, law 1: a source cell's outbox (cap entries) ran out */
    PMCO_ST_TRANS_COPIES = 128   /* transport, law 1: an emitter's copy count met the guard */
};

typedef struct {
    double const_k;       /* constant kernel value, m^3/s                   */
    double additive_beta; /* additive kernel K = beta*(v1+v2), 1/s          */
} pmco_kernel_params;

/* One cell's aero_state_t / env_state_t, as flat arrays (the fields the
 * interface touches, wrf_pmc_driver.F90:1419-1466,1617-1630,611-637). */
typedef struct {
    int32_t n_spec, n_class, n_source, cap, n_bin;
    int32_t max_events, max_appends; /* log capacities shared with the GPU spec */
    const double *density; /* [n_spec] aero_data%density                     */
    const double *v_edge;  /* [n_bin+1] sort-grid volume edges               */
    const double *kmax;    /* [n_bin*n_bin] unweighted kernel upper bounds   */
    double temp, pressure; /* env_state%temp, %pressure                      */
    double *mag;           /* [n_class] awa%weight(1,c)%magnitude (flat)     */
    int32_t *n_part;       /* apa%n_part                                     */
    double *vol;           /* [n_spec][cap] particle%vol                     */
    double *vtot, *mass;   /* [cap] cached sum_s vol, sum_s density*vol      */
    int32_t *wclass;       /* [cap] particle%weight_class (0-based)          */
    int64_t *id;           /* [cap] particle%id                              */
    double *lct, *gct;     /* [cap] least/greatest_create_time               */
    int32_t *n_orig;       /* [n_source][cap] per-source constituent counts  */
    /* aero_state%aero_sorted + %valid_sort (wrf_pmc_trans_aero.F90:2094-2122):
     * kept between calls; every routine that adds, removes or resizes
     * particles other than mc_coag clears sort_valid */
    int32_t pcap;          /* entries of perm: min(65535, cap + cap/8 + min(16 G, 8192)) */
    int32_t *sort_valid;   /* [1]                                            */
    int32_t *perm;         /* [pcap] slot lists, group after group           */
    int32_t *gcnt;         /* [n_class*n_bin] list lengths                   */
    int32_t *goff;         /* [n_class*n_bin + 1] list starts in perm        */
    /* particle%component(:) (wrf_pmc_driver.F90:1421-1437: source_id, create_time per
     * constituent), the first n_comp_max of the list; 0 / NULL: not carried.  A coagulated
     * particle's list is its first candidate's followed by its second's
     * [aero_particle_coagulate, upstream, uncited]; sum_s n_orig is the untruncated length. */
    int32_t n_comp_max;
    int32_t *comp_n;       /* [cap] entries held (<= n_comp_max)             */
    int32_t *comp_src;     /* [cap][n_comp_max] source_id (0-based), -1 past the end */
    double *comp_time;     /* [cap][n_comp_max] create_time, 0 past the end  */
    /* aero_state_prepare_weight_for_add inside aero_state_add_aero_dist_sample [upstream, uncited;
     * the reference hands allow_doubling / allow_halving to every add: wrf_pmc_driver.F90:207-210,
     * 1920-1922, wrf_pmc_trans_aero.F90:1030-1032]: NULL = off (a plain Poisson sample at the
     * class's present weight); else n_part_ideal [n_class] and the two flags */
    const double *prep_n_ideal;
    int32_t prep_doubling, prep_halving;
} pmco_cell;

void pmco_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double pmco_exp_(double x);
double pmco_log_(double x);
double pmco_cbrt_(double x);
double pmco_norminv_(double p);
int32_t pmco_poisson_(double mean, double u);

/* K(v1,m1; v2,m2) [m^3/s] for particle volumes v [m^3] and masses m [kg]. */
double pmco_kernel(int type, double v1, double m1, double v2, double m2,
                   double temp, double pressure, const pmco_kernel_params *kp);

/* Sort-grid tables: n_bin log-spaced radius bins on [r_min, r_max].
 * v_edge[n_bin+1]; v_samp[n_bin*3] = lower edge, geometric centre, upper edge
 * (in volume) of each bin -- the sample points of the K_max estimate. */
void pmco_make_bin_tables(int n_bin, double r_min, double r_max,
                          double *v_edge, double *v_samp);

int pmco_bin_of(const double *v_edge, int n_bin, double v);

/* kmax[b1*n_bin+b2] = (rounded up to a float) max of K over the 3x3 sample volumes of the two bins
 * and the four (rho_min|rho_max)^2 density combinations. */
/* dry deposition of one surface cell (interface/wrf_pmc_dep_aero.F90:154-272) */
int pmco_dry_deposition(pmco_cell *c, double dt, uint64_t seed, uint32_t cell_id, uint32_t step,
                        double aer_res_a, double ustar, double alpha, double big_a,
                        double height, double inv_density, int32_t *unrealistic);
double pmco_dep_velocity_(double diameter, double density, double temp, double rrho, double ustar,
                          double ra, double alpha, double big_a);
float pmco_up_f32(double k);
void pmco_kmax_table(int type, int n_bin, const double *v_samp, double rho_min,
                     double rho_max, double temp, double pressure,
                     const pmco_kernel_params *kp, double *kmax);

/* vtot/mass caches from the species volumes, particles [0, n_part). */
void pmco_refresh_caches(pmco_cell *c);

/* aero_state_sort: lists in ascending slot order, pmco_sort_slack(n) free
 * entries behind a list of n.  Sets sort_valid. */
int pmco_sort_slack(int n, int level);
/* the first level of the ladder whose layout fits pcap entries: N particles, S3 = sum n_g / 8, P
 * populated groups of G */
int pmco_sort_level(int N, int S3, int P, int G, int pcap);
void pmco_sort(pmco_cell *c, int *status);

/* One mc_coag call on one cell.  Returns the status bits.  Sorts first unless
 * sort_valid; the lists it leaves are valid for the next call unless an
 * appended entry did not fit behind its group (then sort_valid is cleared). */
int pmco_mc_coag(pmco_cell *c, int kernel_type, const pmco_kernel_params *kp,
                 double del_t, uint64_t seed, uint32_t cell_id, uint32_t step,
                 int32_t *n_samp, int32_t *n_coag);

/* aero_state_rebalance: per class, double while n < n_ideal/2, halve while
 * n > 2*n_ideal.  next_id is the id counter (module variable next_id,
 * wrf_pmc_driver.F90:1652). */
int pmco_rebalance(pmco_cell *c, const double *n_ideal /*[n_class]*/,
                   int allow_doubling, int allow_halving, uint64_t seed,
                   uint32_t cell_id, uint32_t step, int64_t *next_id);

/* partmc_process observables: tot_num_conc, tot_mass_conc and the two
 * n_dbin histograms over dry diameter (wrf_pmc_driver.F90:783,838-857).
 * i_water < 0 means no water species. */
void pmco_diagnostics(const pmco_cell *c, int i_water, int n_dbin,
                      const double *d_edge /*[n_dbin+1]*/,
                      const double *d_width /*[n_dbin]*/, double *tot_num,
                      double *tot_mass, double *num_dist, double *mass_dist);

/* Synthetic populations: n particles of one cell drawn from a mixture of
 * log-normal modes with fixed volume fractions (the shape of the idealised
 * IC, wrf_pmc_init.F90:1764-1777, and of aero_init_dist.dat). */
typedef struct {
    int32_t n_mode;
    const double *mode_frac;     /* [n_mode] cumulative number fraction     */
    const double *ln_gm_radius;  /* [n_mode] ln of geometric-mean radius    */
    const double *ln_gsd;        /* [n_mode] ln of geometric std deviation  */
    const double *vol_frac;      /* [n_mode][n_spec] species volume fractions */
    const int32_t *mode_class;   /* [n_mode] weight class of the mode       */
    const int32_t *mode_source;  /* [n_mode] source index of the mode       */
} pmco_gen_spec;

void pmco_generate(pmco_cell *c, const pmco_gen_spec *g, int32_t n,
                   uint64_t seed, uint32_t cell_id, int64_t first_id,
                   double create_time);

/* species mass concentrations, number concentration by source, PM1/2.5/10
 * (wrf_pmc_driver.F90:858-932,1153-1157) */
void pmco_diagnostics_ext(const pmco_cell *c, int i_water, double *species_mass /*[n_spec]*/,
                          double *source_num /*[n_source]*/, double *pm /*[3]*/);

/* scenario_update_aero_state (call: wrf_pmc_driver.F90:207-210): emission of
 * new particles from log-normal modes, loss by dilution, gain from the
 * background.  Rates are number concentrations: emission [#/m^3/s],
 * background [#/m^3]; dilution_rate [1/s]. */
typedef struct {
    int32_t n_mode;
    const double *num_conc;      /* [n_mode] rate (emission) or concentration (background) */
    const double *ln_gm_radius;  /* [n_mode] */
    const double *ln_gsd;        /* [n_mode] */
    const double *vol_frac;      /* [n_mode][n_spec] */
    const int32_t *mode_class;   /* [n_mode] */
    const int32_t *mode_source;  /* [n_mode] */
} pmco_aero_dist;

int pmco_scenario_update(pmco_cell *c, double dt, double create_time, uint64_t seed,
                         uint32_t cell_id, uint32_t step, const pmco_aero_dist *emission,
                         double dilution_rate, const pmco_aero_dist *background,
                         int64_t *next_id, int32_t *n_emit, int32_t *n_dil_in,
                         int32_t *n_dil_out);

void pmco_set_transport_law(int law);   /* 0 (default): the library's law; 1: prototype, pmco.c */
int pmco_scenario_update_h(pmco_cell *c, double dt, double create_time, uint64_t seed,
                           uint32_t cell_id, uint32_t step, const pmco_aero_dist *emission,
                           double dilution_rate, const pmco_aero_dist *background,
                           int64_t *next_id, int32_t *n_emit, int32_t *n_dil_in,
                           int32_t *n_dil_out, double height, double old_height);

/* seasalt_emissions (interface/wrf_pmc_driver.F90:1767-1924): the modes of the sea-salt source
 * for option 1 (Gong, the MAM3-matched pair of log-normal modes, :1834-1854) or 2 (Ovadnevaite
 * et al. 2014, five modes, :1886-1912, with the reference's fixed Re = 3.1e6, :1909), and the
 * factor emission_rate_scale (wind_speed**3.41 or 1).  flux [n_modes] in # m^-2 s^-1,
 * gm_radius [m], gsd_log10; returns n_modes (0: unknown option).  Particles are Na and Cl with
 * volume fractions mw_na / (mw_na + mw_cl) and the rest (:1830-1833). */
int pmco_seasalt_modes(int sea_salt_opt, double wind_speed, double *flux, double *gm_radius,
                       double *gsd_log10, double *rate_scale, double *vol_frac_na);
/* one cell's sea-salt source over dt: the emission law of pmco_scenario_update with those modes,
 * rate = flux * rate_scale / height per mode, drawn at RNG kind 2 (so that it is independent of
 * the scenario's own emission in the same step) */
int pmco_seasalt_emissions(pmco_cell *c, double dt, double create_time, uint64_t seed,
                           uint32_t cell_id, uint32_t step, int sea_salt_opt, double wind_speed,
                           double height, int i_na, int i_cl, const int32_t *mode_class,
                           const int32_t *mode_source, int64_t *next_id, int32_t *n_emit);

/* scenario_update_gas_state (call: wrf_pmc_driver.F90:202-205) for one cell: the gas phase
 * gas_state%mix_rat(:) [ppb] gains the emissions and is diluted towards the background
 * [UPSTREAM-UNCITED: PartMC scenario.F90; restated from the published model description,
 * Riemer et al. 2009 eq. 5-6, and the scenario fields the interface fills,
 * wrf_pmc_init.F90:641-657,718-720,789-791,901-919]:
 *   emission [mol m^-2 s^-1] -> ppb: / (p / (R T)) * 1e9, added with the factor
 *       emission_rate_scale * dt / height;
 *   p = exp(-dilution_rate * dt), times old_height / height when the layer grew;
 *   mix_rat = mix_rat * p + background * (1 - p), clipped at zero.
 * The scenario's time series are interpolated first (pmco_interp_1d: constant before the
 * first and after the last point, linear in between). */
double pmco_interp_1d(int n, const double *x, const double *y, int y_stride, double xq);
void pmco_scenario_update_gas(int n_gas, double *mix_rat, double dt, double temp, double pressure,
                              double height, double old_height, const double *emission,
                              double emission_rate_scale, const double *background,
                              double dilution_rate);

/* Stochastic particle transport between the cells of a block:
 * aero_state_multisample + aero_state_sample_particles_by_class
 * (interface/wrf_pmc_trans_aero.F90:2351-2482, 2181-2322), per-particle form.
 * cells[] in WRF order: cell = (j*nz + k)*nx + i.
 *   prob_h [n_cells][n_class][4]  to i-1, i+1, j-1, j+1 (prob_advection + prob_diffusion)
 *   prob_v [n_cells][n_class][nz] to level k' of the same column (prob_vert_diffusion)
 *   cell_volume [n_cells]
 * Lateral boundaries are periodic or closed (a destination outside the block
 * is dropped from the list).  n_moved[c] = particles that left cell c. */
int pmco_transport(pmco_cell *cells, int n_cells, int nx, int nz, int ny, int periodic_x,
                   int periodic_y, const double *prob_h, const double *prob_v,
                   const double *cell_volume, uint64_t seed, uint32_t cell_id_base,
                   uint32_t step, int64_t *next_id, int32_t *n_moved, int32_t *n_arrived);

/* The transport probabilities of one step (wrf_pmc_trans_driver's first half,
 * interface/wrf_pmc_trans.F90:95-156): compute_advect_probs_wrf (:205-286),
 * compute_diffusion_probs (:358-409), compute_vertical_probs (:414-562) with
 * stable_timestep (:629-651), normalize_probs (:692-773), for the nx x nz x ny block
 * cell = (j*nz + k)*nx + i.  The WRF fields are single precision, as in grid%:
 *   alt, xkhh         [ny+2][nz][nx+2]  with one halo cell in x and y (the reference reads
 *                                       i-1, i+1, j-1, j+1 of WRF's memory)
 *   u_flux            [n_class][ny][nz][nx+1]   (class c is the reference's u_flux(:,:,:,c+1))
 *   v_flux            [n_class][ny+1][nz][nx]
 *   w_flux            [n_class][ny][nz+1][nx]
 *   z                 [ny][nz][nx]    z_at_w, exch_h  [ny][nz+1][nx]
 * all_columns: do_uniform, do_scm or doubly periodic (:119-131); otherwise the vertical
 * probabilities exist only for columns with global index 2 .. global_n-1 (1-based), the
 * block's first column being (i_offset, j_offset) of the global domain (0-based).
 * n_parts [n_cells][n_class]: aero_state_total_particles per class; mag likewise.
 * Out: prob_h [n_cells][n_class][4] = prob_advection + prob_diffusion towards i-1, i+1,
 * j-1, j+1; prob_v [n_cells][n_class][nz]; n_substeps [ny][nx] (0 where no vertical
 * probabilities were built); bad [n_cells]: which of the reference's assertions would have
 * stopped the run (1: negative before, 2: negative after, 4: above one, 8: sum >= 1 + 1e-8).
 * The matrix products add their three non-zero terms in ascending inner index. */
typedef struct {
    int nx, nz, ny, n_class;
    double dt, dx, dy;
    int do_vertmix, all_columns, i_offset, j_offset, global_nx, global_ny;
    const float *alt, *xkhh, *u_flux, *v_flux, *w_flux, *z, *z_at_w, *exch_h;
} pmco_trans_fields;
int pmco_transport_probs(const pmco_trans_fields *f, const int32_t *n_parts, const double *mag,
                         double *prob_h, double *prob_v, int32_t *n_substeps, int32_t *bad);

/* The refill of a lateral boundary cell after the transport (interface/
 * wrf_pmc_trans_aero.F90:803-955): a copy of the inner neighbour on outflow
 * (aero_state_dup_all_particles), aero_state_resample (:976-1098) against the background on
 * inflow.  See oracle/pmco.c for the law of each. */
void pmco_boundary_copy(pmco_cell *dst, const pmco_cell *src, int64_t *next_id_dst);
int pmco_boundary_resample(pmco_cell *c, int throw_away, const pmco_aero_dist *bg, double density,
                           double create_time, uint64_t seed, uint32_t cell_id, uint32_t step,
                           int64_t *next_id, int32_t *n_added, int32_t *n_removed);

/* pmco_transport with the reference's open lateral boundaries (real case; interface/
 * wrf_pmc_trans_aero.F90:497-534, 689-801): cells 1 .. n-2 sample as always, a lateral boundary
 * cell only sends inwards with min(prob, 1), a corner nothing, and what is sent to a boundary
 * cell is dropped.  nx, ny >= 3. */
int pmco_transport_lateral(pmco_cell *cells, int n_cells, int nx, int nz, int ny,
                           const double *prob_h, const double *prob_v, const double *cell_volume,
                           uint64_t seed, uint32_t cell_id_base, uint32_t step, int64_t *next_id,
                           int32_t *n_moved, int32_t *n_arrived);

/* trans_aero_preweight (interface/wrf_pmc_trans_aero.F90:1103-1423) with compute_fluxes
 * (:1428-1627): the number concentration every (cell, class) is expected to hold after the
 * transport -- num_conc_old = particles x magnitude, minus prob x num_conc_old towards every
 * destination, plus the same times dest_vol / source_vol at the destination, accumulated in the
 * reference's order (sources in j, k, i order; within a source j-1, j+1, i-1, i+1, then the
 * levels; what crosses a periodic edge gathers in a ghost cell and is added afterwards, y edges
 * before x edges) -- and, where that exceeds twice n_part_ideal, aero_state_scale_weight by
 * n_parts_new / n_part_ideal: the magnitude grows by the ratio and
 * prob_round(n (1 - 1 / ratio)) of the class's n particles, chosen uniformly at random, leave
 * ([upstream, uncited]; here the count from one uniform, the subset by selection sampling in slot
 * order, site SCALE).
 * ideal = periodic_x && periodic_y (:1183-1184); otherwise the lateral boundary cells only send
 * inwards and the corner cells nothing (:1510-1565).  prob_h / prob_v / cell_volume as for
 * pmco_transport; n_ideal [n_cells][n_class].  expected [n_cells][n_class] (= expected_num_conc),
 * n_removed [n_cells]. */
int pmco_transport_preweight(pmco_cell *cells, int n_cells, int nx, int nz, int ny, int periodic_x,
                             int periodic_y, const double *prob_h, const double *prob_v,
                             const double *cell_volume, const double *n_ideal, uint64_t seed,
                             uint32_t cell_id_base, uint32_t step, double *expected,
                             int32_t *n_removed);

/* EXTENSION (no call site on the reference's timestep): water condensation,
 * kappa-Koehler growth ODE per particle, environment fixed over del_t. */
void pmco_condense(pmco_cell *c, const double *kappa /*[n_spec]*/, int i_water,
                   double rel_humid, double del_t, int n_sub);

#ifdef __cplusplus
}
#endif
#endif /* PMCO_H */
