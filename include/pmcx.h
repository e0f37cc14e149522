/*
 * pmcx.h -- C-ABI of the MI355X-native PartMC microphysics path.
 *
 * Drop-in boundary for the per-grid-cell loop body of
 *     partmc_timestep          /root/reference/interface/wrf_pmc_driver.F90:46-290
 * i.e. the calls WRF-PartMC makes per owned cell every chemistry step:
 *     mc_coag(COAG_KERNEL_TYPE_BROWN, env_state, aero_data, aero_state,
 *             del_t, n_samp, n_coag)                        driver:192-194
 *     aero_state_rebalance(aero_state, aero_data, allow_doubling,
 *             allow_halving, ...)                           driver:248-250
 *     aero_info_array_zero                                  driver:251
 * and the partmc_process observables                        driver:783,838-857.
 *
 * The reference has no FFI: Fortran owns aero_state_t (allocatable components,
 * not C-interoperable), so the binding a maintainer adds is the
 * ISO_C_BINDING module wrf-partmc_amd/fortran/pmcx_shim.F90 (see
 * INTEGRATION.md), which gathers aero_state_t into the flat arrays below.
 *
 * Conventions: plain pointers and sizes, host pointers unless a name ends in
 * _dev; every function returns 0 on success or a PMCX_E* code and
 * pmcx_last_error() gives the message (the reference aborts on error,
 * wrf_pmc_trans.F90:713; the Fortran shim maps non-zero to wrf_error_fatal).
 * All cells of a context are batched: one call advances every cell.
 * Indices are 0-based here; the shim converts Fortran's 1-based classes.
 */
#ifndef PMCX_H
#define PMCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMCX_VERSION 1

/* Coagulation kernel ids, numbered as PartMC's COAG_KERNEL_TYPE_*
 * (selected at wrf_pmc_driver.F90:192; objects interface/Makefile:24-30). */
enum {
    PMCX_COAG_KERNEL_TYPE_SEDI = 1,
    PMCX_COAG_KERNEL_TYPE_ADDITIVE = 2,
    PMCX_COAG_KERNEL_TYPE_CONSTANT = 3,
    PMCX_COAG_KERNEL_TYPE_BROWN = 4,
    PMCX_COAG_KERNEL_TYPE_ZERO = 5,
    PMCX_COAG_KERNEL_TYPE_BROWN_SEDI = 101
};

/* error codes */
enum {
    PMCX_OK = 0,
    PMCX_EINVAL = 1,   /* bad argument                                    */
    PMCX_ENOMEM = 2,   /* device or host allocation failed                */
    PMCX_EDEVICE = 3,  /* HIP runtime error / no usable gfx950 device     */
    PMCX_ESTATE = 4,   /* call sequence error (e.g. bounds not computed)  */
    PMCX_ECELL = 5     /* some cell raised a status bit that stops a step */
};

/* per-cell status bits (pmcx_get_coag_counters) */
enum {
    PMCX_ST_EVENT_OVERFLOW = 1,
    PMCX_ST_APPEND_OVERFLOW = 2,
    PMCX_ST_KMAX_EXCEEDED = 4,
    PMCX_ST_OUT_OF_GRID = 8,
    PMCX_ST_CAP_OVERFLOW = 16,
    PMCX_ST_KMAX_RANGE = 32,  /* a particle was created in a bin above the range the cell's
                                 kernel bounds cover (pmcx_set_kernel_bounds_margin)        */
    PMCX_ST_OUTBOX_OVERFLOW = 64, /* transport, law 1: the SOURCE cell's list of entries to send
                                 (cap of them: one per mover, one per destination an emitter
                                 sends copies to) ran out; the first particle whose entries did
                                 not fit and every later one of the cell stayed and sent nothing */
    PMCX_ST_TRANS_COPIES = 128 /* transport, law 1: an emitter's copy count met the guard of 4096
                                 (a weight ratio below about 1e-3 between neighbours)        */
};

typedef struct pmcx_ctx pmcx_ctx;

typedef struct {
    int32_t struct_size;  /* sizeof(pmcx_config), for ABI checking            */
    int32_t device;       /* HIP device ordinal                               */
    int32_t n_cells;      /* cells owned by this context (this GPU's block)   */
    int32_t cap;          /* particle capacity per cell, <= 65280 (16-bit list
                           * entries), whatever the number of weight classes  */
    int32_t n_spec;       /* aero_data_n_spec                                 */
    int32_t n_class;      /* weight classes (n_group is 1, driver:1706)       */
    int32_t n_source;     /* width of the per-particle source histogram       */
    int32_t n_bin;        /* aero_sorted bin grid: bins (<= 128)              */
    double r_min, r_max;  /*   radius range, m                                */
    int32_t max_events;   /* coagulation events per cell per mc_coag call     */
    int32_t max_appends;  /* list appends per cell per call (<= 4096)         */
    int32_t n_dbin;       /* diagnostics diameter grid (100, driver:783)      */
    int32_t i_water;      /* 0-based index of H2O in aero_data, -1 if none    */
    double d_min, d_max;  /*   1e-9, 1e-3 m (driver:783)                      */
    uint32_t cell_id_base;/* global id of local cell 0: RNG key, so results
                             do not depend on how cells are split over GPUs   */
    uint32_t compact_den; /* mc_coag closes a cell's holes (stable compaction) once more
                             than 1/compact_den of its slots are tombstones; default 2.
                             Results never depend on it (the survivors' order is fixed);
                             a full compaction moves every particle, so a small value
                             costs bandwidth, a large one leaves holes for the
                             full-population passes to skip                      */
    double const_k;       /* constant-kernel value, m^3/s                     */
    double additive_beta; /* additive-kernel coefficient, 1/s                 */
    int32_t n_comp_max;   /* particle%component(:) (driver:1421-1437): entries of a particle's
                             list carried on the device, 0..8 (0, the default: only the source
                             histogram n_orig and the least/greatest create times); needs
                             1 <= n_source <= 255; 8 (1 + n_comp_max) bytes per slot and
                             8 max_events bytes per cell                             */
    int32_t pad_;
} pmcx_config;

/* Fills cfg with the reference's defaults (registry.partmc:1-38, driver:783). */
void pmcx_config_default(pmcx_config *cfg);

int pmcx_create(const pmcx_config *cfg, pmcx_ctx **out);
int pmcx_destroy(pmcx_ctx *ctx);
const char *pmcx_last_error(const pmcx_ctx *ctx); /* ctx may be NULL: create errors */
int pmcx_synchronize(pmcx_ctx *ctx);
/* on != 0: the per-step stage calls -- pmcx_set_env, pmcx_transport_probs, pmcx_transport_preweight,
 * pmcx_transport, pmcx_dry_deposition, pmcx_scenario_update_aero_state (pmcx_mc_coag, the gas update
 * and pmcx_rebalance never wait) -- enqueue their uploads, kernels and downloads on the context's
 * stream and return without waiting for the device: one WRF step is then one uninterrupted queue
 * and the host's work (staging the next call's arrays) overlaps the device's.  The caller keeps
 * the input arrays unchanged, and does not read the output arrays, until pmcx_synchronize (or
 * any call that returns values: counters, downloads); with page-locked arrays (pmcx_host_alloc)
 * the transfers themselves are asynchronous too.  What a stage call would have reported after
 * waiting (pmcx_transport_probs' sub-step clamp) is reported by that pmcx_synchronize. */
int pmcx_set_async(pmcx_ctx *ctx, int32_t on);

/* aero_data_t: density(:) [kg/m^3], kappa(:), molec_weight(:)
 * (aero_data.dat columns; wrf_pmc_driver.F90:1420). kappa/molec_weight may be NULL. */
int pmcx_set_aero_data(pmcx_ctx *ctx, const double *density, const double *kappa,
                       const double *molec_weight);

/* env_state_t fields the path reads (producer wrf_pmc_driver.F90:611-637),
 * one value per cell.  rel_humid may be NULL.  The kernel bounds belong to the
 * environment they were computed in: after this call the next pmcx_mc_coag (or an
 * explicit pmcx_update_kernel_bounds) computes them again. */
int pmcx_set_env(pmcx_ctx *ctx, const double *temp, const double *pressure,
                 const double *rel_humid);

/* awa%weight(1,class)%magnitude, flat weights
 * (AERO_STATE_WEIGHT_FLAT_SPECIFIED, wrf_pmc_init.F90:252-253): [n_cells][n_class] */
int pmcx_set_weights(pmcx_ctx *ctx, const double *magnitude);
int pmcx_get_weights(pmcx_ctx *ctx, double *magnitude);

/* aero_state_set_n_part_ideal (wrf_pmc_init.F90:254): [n_cells][n_class] */
int pmcx_set_n_part_ideal(pmcx_ctx *ctx, const double *n_ideal);

/* The allow_doubling / allow_halving the reference hands to every add of particles from a
 * distribution -- scenario_update_aero_state (wrf_pmc_driver.F90:207-210), the sea-salt source
 * (:1920-1922), the refill of an emptied boundary cell (wrf_pmc_trans_aero.F90:1030-1032).
 * on != 0: before a mode's Poisson sample the class's weight is prepared for the add
 * (aero_state_prepare_weight_for_add, [UPSTREAM-UNCITED]): with n the class's particles and
 * n_add the sample's mean at the present weight times the add's characteristic_factor
 * (max(1, min(3600 s, elapsed_time)) / dt in the scenario update and the sea-salt source,
 * wrf_pmc_driver.F90:1794, 1916-1922; 1 in the boundary refill, wrf_pmc_trans_aero.F90:1016,
 * 1031), n + n_add outside [n_ideal / 2, 2 n_ideal]
 * scales the weight by ratio = (n + n_add) / n_ideal (aero_state_scale_weight: ratio > 1 and
 * allow_halving, or an empty class -- exactly prob_round(n (1 - 1 / ratio)) particles leave, a
 * uniformly random subset; ratio < 1 and allow_doubling, or an empty class -- every particle
 * gets prob_round(1 / ratio) - 1 copies under new ids, appended in rank order), and the sample is
 * drawn at the new weight: the class comes to n_ideal, the number concentration stays what it
 * was in expectation.  It is what keeps a boundary cell that was a copy of a much-doubled
 * neighbour from being refilled with thousands of particles at that neighbour's weight.
 * on == 0 (the default): plain Poisson samples at the present weight.  Applies to
 * pmcx_scenario_update_aero_state, pmcx_seasalt_emissions, pmcx_boundary_resample (throw_away
 * branch; the other branch is in-tree code without it, :1035-1095) and pmcx_timestep.
 * PMCX_ESTATE without pmcx_set_n_part_ideal. */
int pmcx_set_weight_preparation(pmcx_ctx *ctx, int32_t on, int32_t allow_doubling, int32_t allow_halving);

/* next_id counters (module variable next_id, wrf_pmc_driver.F90:1652): [n_cells] */
int pmcx_set_next_id(pmcx_ctx *ctx, const int64_t *next_id);
int pmcx_get_next_id(pmcx_ctx *ctx, int64_t *next_id);

/*
 * Upload cells [cell_first, cell_first+n_up).  Particles of cell k occupy
 * [part_offset[k], part_offset[k+1]) of the packed arrays (n_total =
 * part_offset[n_up]):
 *   vol      [n_spec][n_total]    particle%vol(:)           driver:1419-1420
 *   wclass   [n_total]            particle%weight_class-1   driver:1447
 *   id       [n_total]            particle%id               driver:1451
 *   least_ct, greatest_ct [n_total]                         driver:1453-1456
 *   n_orig   [n_source][n_total]  source histogram (replaces component(:),
 *                                 driver:1421-1437); may be NULL
 */
int pmcx_upload_cells(pmcx_ctx *ctx, int32_t cell_first, int32_t n_up,
                      const int64_t *part_offset, const double *vol,
                      const int32_t *wclass, const int64_t *id,
                      const double *least_ct, const double *greatest_ct,
                      const int32_t *n_orig);

/* aero_state_total_particles per cell (driver:143,174): [n_cells] */
int pmcx_get_n_part(pmcx_ctx *ctx, int32_t *n_part);

/* Mirror of upload; part_offset must be the exclusive prefix sum of the
 * counts pmcx_get_n_part returned.  Any output pointer may be NULL. */
int pmcx_download_cells(pmcx_ctx *ctx, int32_t cell_first, int32_t n_dn,
                        const int64_t *part_offset, double *vol, int32_t *wclass,
                        int64_t *id, double *least_ct, double *greatest_ct,
                        int32_t *n_orig);

/*
 * particle%component(:)%source_id / %create_time (wrf_pmc_driver.F90:1421-1437) of the cells
 * pmcx_upload_cells has just laid out (same part_offset; call it right behind that upload), for
 * a context created with n_comp_max > 0:
 *   comp_len    [n_total]              entries given for the particle (those beyond n_comp_max
 *                                      are not carried)
 *   comp_source [n_total][n_comp_max]  source_id - 1
 *   comp_time   [n_total][n_comp_max]  create_time
 * Without this call an uploaded particle's list is rebuilt from its histogram: n_orig(s) entries
 * per source in source order, the first created at least_ct, the others at greatest_ct (exact for
 * a particle of one constituent).  From then on the lists follow the particles through every
 * entry point: a new particle has one entry (its mode's source, the create time of the call), a
 * copy its original's list, and a coagulated particle the list of its first candidate followed by
 * that of its second [aero_particle_coagulate: upstream, uncited], cut at n_comp_max -- the
 * untruncated length stays available as sum_s n_orig(s).  The download returns, per particle, the
 * entries held (comp_len <= n_comp_max), -1 / 0 past the end.
 */
int pmcx_upload_components(pmcx_ctx *ctx, int32_t cell_first, int32_t n_up,
                           const int64_t *part_offset, const int32_t *comp_len,
                           const int32_t *comp_source, const double *comp_time);
int pmcx_download_components(pmcx_ctx *ctx, int32_t cell_first, int32_t n_dn,
                             const int64_t *part_offset, int32_t *comp_len,
                             int32_t *comp_source, double *comp_time);

/* est_k_minmax_binned_unweighted: (re)compute every cell's kernel upper
 * bounds for kernel_type from the current env (PartMC caches this per
 * aero_state, coag_kernel_bounds_valid). */
int pmcx_update_kernel_bounds(pmcx_ctx *ctx, int32_t kernel_type);

/* Without that explicit call pmcx_mc_coag keeps the bounds current by itself, after its sort:
 * a cell gets bounds for its populated bins plus `margin` bins above them (default 8: a
 * particle 250 times the volume of the cell's largest) whenever the kernel type, its
 * temperature or pressure (pmcx_set_env), or -- through anything but coagulation -- its
 * population changed, and every 16th call for the growth coagulation itself causes; the rest
 * of its table is stale.  Should a call create a particle above the covered range the cell's
 * status carries PMCX_ST_KMAX_RANGE (the pairs of that bin were sampled with a stale bound).
 * margin < 0: whole tables, computed when the kernel type or the environment changes
 * (est_k_minmax_binned_unweighted as upstream runs it; 9 ms per 1.2e5 cells). */
int pmcx_set_kernel_bounds_margin(pmcx_ctx *ctx, int32_t margin);

/*
 * mc_coag on every cell (wrf_pmc_driver.F90:192-194).  Asynchronous on the
 * context's stream with one weight class; with several the call waits for the
 * device once or more (its pair sampler works in rounds and reads back whether a
 * cell is left).  seed = grid%random_seed (wrf_pmc_init.F90:131);
 * step = chemistry-step index; together with the global cell id they key the
 * counter-based RNG.  The species volumes of coagulated particles are summed
 * behind the sampler, in event order, from the log of its events (n_spec > 1).
 */
int pmcx_mc_coag(pmcx_ctx *ctx, int32_t kernel_type, double del_t, uint64_t seed,
                 uint32_t step);

/* tot_n_samp, tot_n_coag of the last pmcx_mc_coag and the status bits,
 * each [n_cells]; any pointer may be NULL.  Synchronises. */
int pmcx_get_coag_counters(pmcx_ctx *ctx, int32_t *n_samp, int32_t *n_coag,
                           int32_t *status);

/* Sums over cells of the last call's counters (device-side reduction). */
int pmcx_get_coag_totals(pmcx_ctx *ctx, int64_t *n_samp, int64_t *n_coag,
                         int64_t *n_part, int32_t *status_or);

/* Asynchronous form: add the last call's sums to running totals kept on the
 * device (no host synchronisation per step); read (and optionally reset) them later. */
int pmcx_accumulate_totals(pmcx_ctx *ctx);
int pmcx_read_accumulated(pmcx_ctx *ctx, int64_t *n_samp, int64_t *n_coag, int64_t *n_part,
                          int32_t *status_or, int32_t reset);

/* aero_state_rebalance on every cell (wrf_pmc_driver.F90:248-250). */
int pmcx_rebalance(pmcx_ctx *ctx, int32_t allow_doubling, int32_t allow_halving,
                   uint64_t seed, uint32_t step);

/*
 * scenario_update_aero_state on every cell (wrf_pmc_driver.F90:207-210):
 * emission of new particles from log-normal modes, loss by dilution, gain
 * from the background.  An aero_dist is n_mode log-normal modes
 * (gm_radius [m], gsd_log10, vol_frac[n_mode][n_spec], weight class, source)
 * with a number concentration per cell and mode: emission rate [#/m^3/s] or
 * background concentration [#/m^3], both [n_cells][n_mode].  dilution_rate
 * [1/s] is [n_cells] (NULL: no dilution).  New particles get create_time =
 * elapsed_time and ids from the cell's next_id.  Counters are [n_cells], any
 * may be NULL.  Either dist may be NULL.  Synchronises.
 */
typedef struct {
    int32_t n_mode;
    const double *num_conc;    /* [n_cells][n_mode]; negative: the cell's distribution has no such
                                * mode (cells bring different numbers of modes, or none): nothing is
                                * drawn for it and no weight is prepared (pmcx_set_weight_preparation) */
    const double *gm_radius;   /* [n_mode] */
    const double *gsd_log10;   /* [n_mode] */
    const double *vol_frac;    /* [n_mode][n_spec] */
    const int32_t *mode_class; /* [n_mode] */
    const int32_t *mode_source;/* [n_mode] */
    int32_t per_cell;          /* 0: as above, one set of mode parameters for every cell.
                                * 1: every cell has its own -- gm_radius, gsd_log10, mode_class,
                                * mode_source [n_cells][n_mode], vol_frac [n_cells][n_mode][n_spec] -- as
                                * scenario(i,k,j)%aero_emission / %aero_background hold them: read from
                                * each cell's own file (wrf_pmc_init.F90:689-693,705-706,768-786;
                                * 886-892,927-943).  For pmcx_boundary_resample: [n_list][..]. */
} pmcx_aero_dist;

/* The mode parameters of scenario(i,k,j)%aero_emission (which = 0) or %aero_background (1) at
 * their current time level, kept on the device: everything of `dist` but num_conc.  Afterwards a
 * pmcx_aero_dist handed to pmcx_scenario_update_aero_state / pmcx_timestep with gm_radius == NULL
 * stands for these modes and brings only its num_conc [n_cells][n_mode] -- what changes from step
 * to step (rate_scale / height) while the modes hold until a cell's series moves to its next time
 * point (aero_dist_interp_1d keeps a dist between two points [upstream, uncited]).  dist == NULL:
 * forgotten. */
int pmcx_set_aero_scenario_modes(pmcx_ctx *ctx, int32_t which, const pmcx_aero_dist *dist);

/* height / old_height [n_cells] = env_state%height / old_env_state%height (the reference hands both
 * states to the update, wrf_pmc_driver.F90:207-209; NULL: unchanged): where the layer grew the
 * dilution's p = exp(-rate dt) is multiplied by old_height / height, as in the gas update --
 * entrainment of background air; a particle is lost with probability 1 - p and the background
 * is sampled with that factor. */
int pmcx_scenario_update_aero_state(pmcx_ctx *ctx, double dt, double elapsed_time, uint64_t seed,
                                    uint32_t step, const pmcx_aero_dist *emission,
                                    const double *dilution_rate,
                                    const pmcx_aero_dist *background, int32_t *n_emit,
                                    int32_t *n_dil_in, int32_t *n_dil_out,
                                    const double *height, const double *old_height);

/* seasalt_emissions (interface/wrf_pmc_driver.F90:212-223, 1767-1924) on every cell with
 * ocean_surface[c] != 0 (the reference: lowest level, grid%xland > 1.5): log-normal modes of Na
 * and Cl (volume fractions mw_na / (mw_na + mw_cl) and the rest, :1830-1833) --
 *   sea_salt_opt 1  Gong, the two MAM3-matched modes, number fluxes x wind_speed**3.41
 *                   (:1819-1822, 1834-1854)
 *   sea_salt_opt 2  Ovadnevaite et al. 2014, five modes, F_i = coef (Re - offset)**expon with the
 *                   reference's fixed Re = 3.1e6 (:1800-1805, 1886-1912)
 * -- sampled like an emission: Poisson(flux * rate_scale * dt / height / magnitude) new particles
 * per mode (:1917-1924).  wind_speed = sqrt(u10^2 + v10^2) [n_cells] (driver:215-216), height =
 * env_state%height [n_cells]; i_na, i_cl: 0-based species indices; mode_class / mode_source
 * [n_modes] (the reference looks "sea_salt_NN" up in aero_data%weight_class_name / %source_name;
 * NULL: class 0, source 0).  n_emit [n_cells] and rate [n_cells][n_modes] (the emission rate
 * used, # m^-3 s^-1) are outputs and may be NULL.  Synchronises. */
int pmcx_seasalt_emissions(pmcx_ctx *ctx, double dt, double elapsed_time, uint64_t seed, uint32_t step,
                           int32_t sea_salt_opt, int32_t i_na, int32_t i_cl,
                           const int32_t *ocean_surface, const double *wind_speed,
                           const double *height, const int32_t *mode_class,
                           const int32_t *mode_source, int32_t *n_emit, double *rate);

/*
 * The transport probabilities of one step, built on the device from WRF's fields and the
 * particle counts where they are (wrf_pmc_trans_driver's first half,
 * interface/wrf_pmc_trans.F90:95-156): compute_advect_probs_wrf (:205-286),
 * compute_diffusion_probs (:358-409), compute_vertical_probs (:414-562) with stable_timestep
 * (:629-651), normalize_probs (:692-773).  Host arrays, single precision as grid% holds them,
 * for the nx x nz x ny block, cell = (j*nz + k)*nx + i:
 *   alt, xkhh       [ny+2][nz][nx+2]   one halo cell in x and y (grid%alt, grid%xkhh at i-1 .. i+1,
 *                                      j-1 .. j+1 of WRF's memory)
 *   u_flux          [n_class][ny][nz][nx+1]   (class c: grid%u_flux(:,:,:,c+2) in Fortran counting)
 *   v_flux          [n_class][ny+1][nz][nx]
 *   w_flux          [n_class][ny][nz+1][nx]
 *   z               [ny][nz][nx];   z_at_w, exch_h  [ny][nz+1][nx]
 * dt, dx, dy: grid%dt, %dx, %dy; do_vertmix: config_flags%vertmix_onoff; all_columns: do_uniform,
 * do_scm or doubly periodic (:119-131) -- otherwise vertical probabilities exist only for the
 * columns 2 .. global_n-1 (1-based) of the global domain, the block's first column being
 * (i_offset, j_offset) (0-based).  aero_state_total_particles per class and the weight
 * magnitudes are read from the context.
 * The result stays on the device for the NEXT pmcx_transport / pmcx_transport_begin called with
 * prob_h == prob_v == NULL (used once: the probabilities describe the populations as they were).
 * Optional host copies: prob_h [n_cells][n_class][4], prob_v [n_cells][n_class][nz], n_substeps
 * [ny][nx] (stable_timestep's count; 0: column skipped), bad [n_cells] (the reference's
 * assertions 414897882 / 184874367 / 184874369 / 414897881 as bits 1 / 2 / 4 / 8; bit 16: more
 * than 10^4 sub-steps asked for -- the call then fails with PMCX_EINVAL and no probabilities are
 * left for the transport: the reference would loop that often on the CPU), n_class_part
 * [n_cells][n_class].  Synchronises.
 */
typedef struct pmcx_trans_fields {
    int32_t nx, nz, ny;
    int32_t do_vertmix, all_columns;
    int32_t i_offset, j_offset, global_nx, global_ny;
    double dt, dx, dy;
    const float *alt, *xkhh, *u_flux, *v_flux, *w_flux, *z, *z_at_w, *exch_h;
} pmcx_trans_fields;
int pmcx_transport_probs(pmcx_ctx *ctx, const pmcx_trans_fields *f, double *prob_h, double *prob_v,
                         int32_t *n_substeps, int32_t *bad, int32_t *n_class_part);

/* Open lateral boundaries for pmcx_transport, the reference's real case with neither direction
 * periodic (interface/wrf_pmc_trans_aero.F90:497-534, 689-801): cells 1 .. n-2 sample as always
 * (also towards the boundary cells); a lateral boundary cell only sends inwards, with
 * min(prob_advection + prob_diffusion, 1); a corner cell sends nothing; what is sent to a boundary
 * cell leaves the domain.  The boundary cells are then refilled by pmcx_boundary_copy (outflow) or
 * pmcx_boundary_resample (inflow), as :803-955 does.  At least 3 x 3 columns.  For a slab of a
 * domain cut in y (pmcx_transport_begin .. _finish): j_global0 = the domain row its row 0 stands
 * for (-1 for the first rank's lower ghost row), global_ny = the domain's rows; 0, 0: the block is
 * the domain (pmcx_transport). */
int pmcx_set_transport_lateral(pmcx_ctx *ctx, int32_t on, int32_t j_global0, int32_t global_ny);

/* The refill of the lateral boundary cells after the transport (interface/
 * wrf_pmc_trans_aero.F90:803-955).
 * pmcx_boundary_copy: outflow (:829-831) -- cell to_cell[q] becomes a copy of from_cell[q]
 *   (aero_state_dup_all_particles [upstream, uncited]: the same particles in the same order under
 *   new ids, the same weights).
 * pmcx_boundary_resample: inflow (:832-845) -- aero_state_resample (:976-1098) of the listed cells
 *   against the background (num_conc [n_list][n_mode]: number mixing ratios; density [n_list] =
 *   1 / inverse_density; create_time: 0 at the reference's call sites).  throw_away[q] != 0: the
 *   cell is emptied and sampled anew; else every class is brought to a Poisson-distributed target
 *   count -- short of it, new particles from its modes (only if the shortfall is more than a tenth
 *   of the target, :1066-1079); over it, a uniformly random subset of exactly the excess leaves.
 *   n_added / n_removed [n_list] may be NULL.  Both synchronise. */
int pmcx_boundary_copy(pmcx_ctx *ctx, int32_t n_pairs, const int32_t *from_cell, const int32_t *to_cell);
int pmcx_boundary_resample(pmcx_ctx *ctx, int32_t n_list, const int32_t *cell, const int32_t *throw_away,
                           const pmcx_aero_dist *background, const double *density, double create_time,
                           uint64_t seed, uint32_t step, int32_t *n_added, int32_t *n_removed);

/* trans_aero_preweight (interface/wrf_pmc_trans_aero.F90:1103-1423, compute_fluxes :1428-1627),
 * the step in front of the sampling: the number concentration every (cell, class) is expected
 * to hold after this transport (particles x magnitude, less prob x that towards every
 * destination, plus the same x dest_vol / source_vol where it arrives; accumulated in the
 * reference's order) and, where that is more than twice n_part_ideal, aero_state_scale_weight
 * by n_parts_new / n_part_ideal (the magnitude grows by that ratio and exactly
 * prob_round(n (1 - n_part_ideal / n_parts_new)) of the class's n particles, a uniformly random
 * subset, leave).  periodic_x && periodic_y is the reference's "ideal"
 * case; otherwise lateral boundary cells only send inwards and corners nothing.  prob_h ==
 * prob_v == NULL: the probabilities pmcx_transport_probs built (they stay for the transport).
 * Outputs (may be NULL): expected_num_conc [n_cells][n_class], n_removed [n_cells].
 * One block of cells (across ranks: pmcx_transport_preweight_begin / _finish below).
 * Synchronises. */
int pmcx_transport_preweight(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                             int32_t periodic_y, const double *prob_h, const double *prob_v,
                             const double *cell_volume, uint64_t seed, uint32_t step,
                             double *expected_num_conc, int32_t *n_removed);

/* pmcx_transport_preweight for a SLAB of a domain cut in y across ranks (rows 0 and ny - 1 of the
 * block are ghosts of the neighbours' rows, as for pmcx_transport_begin .. _finish), with the
 * whole domain's result bit for bit: what a rank's first and last real rows send across the cut
 * travels as ready terms (flux x dest_vol / source_vol per cell and class), and the receiver adds
 * them where the whole domain's order of summation has them.
 *   _begin   j_global0: the domain row the block's row 0 stands for (-1 for the first rank's lower
 *            ghost); global_ny: rows of the whole domain; prob_h / prob_v [n_cells][..] for every
 *            local cell (ghost rows: unused) or NULL for the ones pmcx_transport_probs built -- given
 *            ones stay on the device for _finish and for a pmcx_transport_begin(.., NULL, NULL, ..)
 *            that follows; edge_terms [2][nx nz][n_class] out: side 0 for the rank below, 1 above
 *   _finish  the neighbours' terms (from below: their side 1; from above: their side 0; NULL: no
 *            neighbour); *_is_edge: that neighbour lies across the domain's periodic edge; then
 *            the scaling.  Outputs as pmcx_transport_preweight ([n_cells], ghost rows zero). */
int pmcx_transport_preweight_begin(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                                   int32_t periodic_y, int32_t j_global0, int32_t global_ny,
                                   const double *prob_h, const double *prob_v, const double *cell_volume,
                                   double *edge_terms);
/* A block cut in x as well (pmcx_set_transport_cuts): pmcx_transport_preweight_begin's edge_terms then
 * hold [2][nx nz][n_class] (towards row 0, towards row ny - 1) followed by [2][ny nz][n_class] (what the
 * first real column sends west, the last sends east), and the terms of the western / eastern neighbours
 * are handed over here, between _begin and _finish ([ny nz][n_class] each, NULL: no neighbour;
 * *_is_edge: that side is the domain's periodic edge, whose term joins the sum last). */
int pmcx_transport_preweight_x_terms(pmcx_ctx *ctx, const double *terms_from_west, const double *terms_from_east,
                                     int32_t west_is_edge, int32_t east_is_edge);
int pmcx_transport_preweight_finish(pmcx_ctx *ctx, const double *terms_from_below, const double *terms_from_above,
                                    int32_t below_is_edge, int32_t above_is_edge, uint64_t seed, uint32_t step,
                                    double *expected_num_conc, int32_t *n_removed);

/* The same transport for a SLAB of a domain decomposed in y across ranks (one process per
 * GPU; wrf_pmc_trans_aero.F90 exchanges the movers between MPI patches).  The block carries
 * one ghost row below its first real row (j = 0) and one above its last (j = ny - 1); the
 * ghost cells hold no particles between steps, but their weights (pmcx_set_weights) and
 * cell volumes must be those of the neighbouring rank's boundary rows, and
 * cell_id_base must make cell_id_base + local index the global cell id (RNG key).
 *   pmcx_transport_begin        every cell draws its movers (no periodicity in y: close a side
 *                               that has no neighbour by zero probabilities towards it)
 *   pmcx_transport_export_size  bytes of the movers of the boundary row that leave through
 *                               side 0 (row 1 -> ghost row 0) or side 1 (row ny-2 -> row ny-1)
 *   pmcx_transport_export       packs them into a DEVICE buffer of that size (a header of
 *                               per-cell counts, then fixed-size records in source-cell and
 *                               outbox order); ship it with RCCL / torch.distributed
 *   pmcx_transport_import       lays a neighbour's buffer out in ghost row 0 (side 0) or ny-1
 *                               (side 1) exactly as that neighbour's boundary row looks to a
 *                               destination (NULL / 0 bytes: no neighbour)
 *   pmcx_transport_finish       arrivals are pulled in the fixed source order (ghost rows
 *                               included), movers removed, ghost rows emptied
 * The result on the real rows is bit for bit the one of pmcx_transport on the whole domain
 * (tests/test_gpu_parity.py::test_transport_slabs_match_whole_domain).
 * One shape is excluded: a PERIODIC direction of the whole domain two cells wide must not be
 * cut.  Both neighbours of a cell are then one cell, whose movers pmcx_transport takes as a single
 * sequence while the exchange delivers them as two (from below and from above): the arrival
 * order would differ.  The host drivers refuse it (CellBlock.transport_distributed[_2d]);
 * pmcx_transport_import does where it can see it (cut in x with global_nx == 2). */
/* A block cut in x as well (the reference decomposes in x and y, wrf_pmc_trans_aero.F90:148-273):
 * cut_x != 0 says columns 0 and nx_local - 1 of the block are ghosts of the ranks to the west and
 * east, local column 0 standing for the domain's column i_global0 (-1 on the west-most rank of a
 * periodic domain means its last column) of global_nx.  The exchange then has four sides -- 0 below,
 * 1 above, 2 west, 3 east -- in pmcx_transport_export_size / _export / _import; a mover goes to one
 * of its four neighbours or a level of its column, so corners never carry anything.  Because a row
 * of such a block is shorter than a row of the domain, the global id of local cell c -- the key of
 * every random draw, in every entry point -- becomes cell_id_base + (c / nx_local) global_nx +
 * c % nx_local, cell_id_base being the global id of local cell 0 (modulo 2^32).  cut_x = 0: back to
 * slabs along y (global id = cell_id_base + c).  Set once, before anything draws. */
int pmcx_set_transport_cuts(pmcx_ctx *ctx, int32_t cut_x, int32_t nx_local, int32_t i_global0,
                            int32_t global_nx);
/* plain device buffers on the context's GPU for the halo exchange (callers that already hold
 * device memory, e.g. torch tensors handed to RCCL, pass their own pointers instead) */
int pmcx_device_alloc(pmcx_ctx *ctx, int64_t n_bytes, void **dev_ptr);
int pmcx_device_free(pmcx_ctx *ctx, void *dev_ptr);
/* kind: 0 device -> device, 1 host -> device, 2 device -> host; synchronises */
int pmcx_device_copy(pmcx_ctx *ctx, void *dst, const void *src, int64_t n_bytes, int32_t kind);
/* Page-locked host memory for the arrays a caller hands to the per-step entry points (the fields of
 * pmcx_transport_probs, the resistances of pmcx_dry_deposition, emission rates, the counters that
 * come back): from such a buffer an upload is one DMA at the link's rate that overlaps whatever
 * the device is doing, from pageable memory the runtime pins or stages page by page while the
 * caller waits (1.5 ms of a 6.6 ms step at 122 880 cells).  The Fortran shim keeps its
 * per-step staging arrays in it.  Any host pointer is accepted everywhere; this only makes it fast. */
int pmcx_host_alloc(pmcx_ctx *ctx, int64_t n_bytes, void **host_ptr);
int pmcx_host_free(pmcx_ctx *ctx, void *host_ptr);
int pmcx_transport_begin(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                         const double *prob_h, const double *prob_v, const double *cell_volume,
                         uint64_t seed, uint32_t step);
int pmcx_transport_export_size(pmcx_ctx *ctx, int32_t side, int64_t *n_bytes);
int pmcx_transport_export(pmcx_ctx *ctx, int32_t side, void *dev_buf);
int pmcx_transport_import(pmcx_ctx *ctx, int32_t side, const void *dev_buf, int64_t n_bytes);
int pmcx_transport_finish(pmcx_ctx *ctx, int32_t *n_moved, int32_t *n_arrived);

/* Dry deposition of aerosol particles (dep_aero_dry_driver / dry_dep_aero_state,
 * interface/wrf_pmc_dep_aero.F90:28-192): on every cell with surface[c] != 0
 * (NULL: every cell; the reference applies it to the lowest level of interior
 * columns, :131-151) each particle is removed with probability (dt / height) vd,
 * vd the Zhang et al. (2001) deposition velocity of the whole particle
 * (:196-234, :265-420) from the aerodynamic resistance aer_res_a [s/m], the
 * friction velocity ustar [m/s], the land-use impaction parameter alpha and the
 * collector radius big_a = A(season, land use) [m] (tables at :94-130; the
 * land-use gamma is overwritten by the reference, :325, and not an input),
 * env_state%height [m], env_state%inverse_density [m^3/kg] and the cell's
 * temperature (pmcx_set_env).  One uniform per particle, keyed by
 * (seed, global cell, step, rank of the particle among the live ones).
 * n_removed, unrealistic ("Unrealistic deposition": a probability outside (0,1),
 * :181-183) are [n_cells] outputs and may be NULL.  Synchronises. */
int pmcx_dry_deposition(pmcx_ctx *ctx, double dt, uint64_t seed, uint32_t step,
                        const int32_t *surface, const double *aer_res_a, const double *ustar,
                        const double *alpha, const double *big_a, const double *height,
                        const double *inv_density, int32_t *n_removed, int32_t *unrealistic);

/*
 * Stochastic particle transport between the cells of this context (SURVEY
 * section 8, row f2): aero_state_multisample + aero_state_sample_particles_by_class
 * (interface/wrf_pmc_trans_aero.F90:2351-2482, 2181-2322) in per-particle form.
 * The context's cells form an nx x nz x ny block in WRF's (i,k,j) order,
 * cell = (j*nz + k)*nx + i.  prob_h [n_cells][n_class][4]: probability to move to
 * i-1, i+1, j-1, j+1 (prob_advection + prob_diffusion, :2419-2445); prob_v
 * [n_cells][n_class][nz]: to level k' of the column (prob_vert_diffusion,
 * :2446-2460; the own level is ignored); cell_volume [n_cells] (the
 * dz_source/dz_dest weight factor, :2214-2218).  Lateral boundaries are periodic
 * or closed (a periodic direction one cell wide counts as closed -- the cell would be its
 * own neighbour -- here and in pmcx_transport_preweight); exchange with other contexts (other GPUs) is not part of this call.
 * prob_h == prob_v == NULL: the probabilities pmcx_transport_probs built (PMCX_ESTATE if none).
 * n_moved / n_arrived: [n_cells], may be NULL.  Synchronises.
 */
/* How a particle that goes where particles stand for less (r = weight there / weight here < 1) turns
 * into 1 / r arrivals.  The reference (wrf_pmc_trans_aero.F90:2236-2300) draws Poisson(n p / r) picks
 * with replacement; every pick adds a copy and removes the picked particle with probability r.
 *   law 0            a mover carries 1 + geometric copies (mean 1 / r): the reference's mean, its
 *                    variance of the arrivals times (2 - r - p) / r (the library's law until the
 *                    end of round 3)
 *   law 1 (default)  the reference's law by Poisson thinning of its picks: the removing picks are
 *                    the movers, each arriving ONCE with its id; the non-removing picks, rate
 *                    mu_d = p_d (1 - r_d) / r_d, are copies from particles that STAY -- behind the
 *                    movers' intervals the placing uniform has one more of width 1 - exp(-sum mu):
 *                    an emitter draws a zero-truncated Poisson(sum mu) number of copies and deals
 *                    them to the r < 1 destinations in proportion mu, every copy under a new id.
 *                    Mean n p / r, variance n p / r: the reference's.  Where the two do not fit
 *                    side by side (sum p + 1 - exp(-sum mu) > 1: from r of about 0.4 at sum p = 0.5)
 *                    the emitters' interval is [exp(-sum mu), 1) and overlaps the movers': a
 *                    particle in the overlap sends its copies AND moves (the reference picks a
 *                    particle without removal before the pick that removes it), so the mean stays
 *                    n p / r at every ratio.  Above sum mu = 32 the count is the plain Poisson's.
 * Applies to pmcx_transport and pmcx_transport_begin ... _finish. */
int pmcx_set_transport_law(pmcx_ctx *ctx, int32_t law);

int pmcx_transport(pmcx_ctx *ctx, int32_t nx, int32_t nz, int32_t ny, int32_t periodic_x,
                   int32_t periodic_y, const double *prob_h, const double *prob_v,
                   const double *cell_volume, uint64_t seed, uint32_t step, int32_t *n_moved,
                   int32_t *n_arrived);

/*
 * EXTENSION -- water condensation as a per-particle growth-ODE sub-step over
 * del_t at each cell's (temp, pressure, rel_humid): kappa-Koehler equilibrium,
 * Seinfeld & Pandis eq. 17.70 growth law, n_sub sub-steps.  The reference
 * links condense.o but its timestep never calls the growth ODE
 * (condense_equilib_particles only, inside optics, wrf_pmc_driver.F90:1181,1201);
 * the BASELINE north_star names it, so it is provided as a labelled extension
 * with its own oracle (parity unpinned).  Needs kappa (pmcx_set_aero_data),
 * rel_humid (pmcx_set_env) and cfg.i_water >= 0.
 */
int pmcx_condense(pmcx_ctx *ctx, double del_t, int32_t n_sub);

/* partmc_process observables (wrf_pmc_driver.F90:838-857):
 * tot_num_conc[n_cells], tot_mass_conc[n_cells],
 * num_dist[n_cells][n_dbin], mass_dist[n_cells][n_dbin]. */
int pmcx_diagnostics(pmcx_ctx *ctx, double *tot_num_conc, double *tot_mass_conc,
                     double *num_dist, double *mass_dist);

/* More partmc_process observables:
 *   species_mass_conc [n_cells][n_spec]   sum(masses(include=species) * num_concs), driver:864-932
 *   num_conc_by_source [n_cells][n_source] aero_state_num_concs_by_source, driver:858-861
 *   pm_mass_conc [n_cells][3]             PM1, PM2.5, PM10 dry mass, driver:1153-1157
 *   n_parts [n_cells]                     aero_state_n_part, driver:1137
 * any pointer may be NULL. */
int pmcx_diagnostics_ext(pmcx_ctx *ctx, double *species_mass_conc, double *num_conc_by_source,
                         double *pm_mass_conc, int32_t *n_parts);

/*
 * Synthetic populations generated on the device (shape of
 * aero_state_add_aero_dist_sample for log-normal modes,
 * wrf_pmc_init.F90:1063,1764-1777): every cell gets n_part particles from
 * the mode mixture.  mode_frac = number fractions; gm_radius in m;
 * gsd_log10 = log10 of the geometric std dev; vol_frac[n_mode][n_spec];
 * ids are cell_global_id*2^32 + slot.
 */
int pmcx_generate_lognormal(pmcx_ctx *ctx, int32_t n_part, int32_t n_mode,
                            const double *mode_frac, const double *gm_radius,
                            const double *gsd_log10, const double *vol_frac,
                            const int32_t *mode_class, const int32_t *mode_source,
                            uint64_t seed, double create_time);

/* gas_state_t%mix_rat(:) [ppb] of every cell (wrf_pmc_driver.F90:1698,1726-1762):
 * [n_cells][n_gas].  The first call fixes n_gas (gas_data_n_spec: 77 in the reference's
 * gas_data.dat, wrf_pmc_init.F90:1492); later calls must repeat it. */
int pmcx_set_gas_state(pmcx_ctx *ctx, int32_t n_gas, const double *mix_rat);
int pmcx_get_gas_state(pmcx_ctx *ctx, double *mix_rat);

/* The gas half of scenario_t as the interface fills it (wrf_pmc_init.F90:646-657, 718-720,
 * 789-791, 901-919), kept on the device:
 *   gas_emission_time [n_time_e], gas_emission_rate_scale [n_time_e],
 *   gas_emission [n_cells][n_time_e][n_gas]            mol m^-2 s^-1
 *   gas_dilution_time [n_time_d], gas_dilution_rate [n_cells][n_time_d]   1/s
 *   gas_background [n_cells][n_time_d][n_gas]          ppb
 * n_time_e = 0: no emissions; n_time_d = 0: no dilution (the arrays of that half may be NULL).
 * Needs pmcx_set_gas_state first (n_gas). */
int pmcx_set_gas_scenario(pmcx_ctx *ctx, int32_t n_time_e, const double *emission_time,
                          const double *emission_rate_scale, const double *emission,
                          int32_t n_time_d, const double *dilution_time,
                          const double *dilution_rate, const double *background);

/* scenario_update_gas_state on every cell (wrf_pmc_driver.F90:202-205): the scenario's series
 * interpolated at elapsed_time (constant outside their range, linear inside); emissions
 * converted to ppb with the cell's air molar density p / (R T) (pmcx_set_env) and added with
 * the factor rate_scale dt / height; dilution towards the background with
 * p = exp(-rate dt), times old_height / height where the layer grew.  height [n_cells] =
 * env_state%height, old_height = old_env_state%height (NULL: unchanged).  Asynchronous. */
int pmcx_scenario_update_gas_state(pmcx_ctx *ctx, double dt, double elapsed_time,
                                   const double *height, const double *old_height);

/* The cells the loop nest of partmc_timestep covers (wrf_pmc_driver.F90:157-171): every owned cell
 * when the domain is doubly periodic, otherwise the interior 2 .. global_n-1 in x and y.  in_loop
 * [n_cells], 0: pmcx_timestep (mc_coag, both scenario updates, sea salt, rebalance) and pmcx_rebalance
 * (trans_aero's rebalance runs over the same cells, wrf_pmc_trans_aero.F90:775-801) leave the cell
 * alone; its counters read zero.  NULL: every cell again (the default).  pmcx_mc_coag and the
 * scenario updates called on their own honour it too. */
int pmcx_set_loop_cells(pmcx_ctx *ctx, const int32_t *in_loop);

/* The largest cap a configuration admits: 65280 (slots are 16-bit list entries; a multiple of 64),
 * whatever n_class -- until round 4 the room laid out behind every group, empty ones included, tied it
 * to 16 n_class n_bin (12 742 slots with 40 classes x 80 bins).  0: n_class n_bin beyond 65535. */
int32_t pmcx_max_cap(const pmcx_config *cfg);

/* Growing particle arrays.  PartMC's aero_particle_array grows when a particle is added to a full one
 * (behind aero_state_add_particle, wrf_pmc_trans_aero.F90:777-804, 2284); a block here has cap slots
 * per cell.  pmcx_grow re-pitches every per-slot array to new_cap slots per cell (rounded up to 64, at
 * most 65280; a smaller or equal one: nothing happens): particles, lists, sort keys, component lists
 * and an open transport's outboxes keep their places -- no result depends on cap -- and the old
 * arrays are freed; the block and its larger copy must both fit in device memory while it grows
 * (PMCX_ENOMEM otherwise, the block unchanged).  Synchronises.
 *
 * pmcx_set_auto_grow(on, fill, factor): the library grows the block by itself --
 *   * pmcx_transport / pmcx_transport_finish: every cell's arrivals are counted by its sources
 *     before anything is written to it; if some cell would not hold its particles plus its arrivals
 *     the block grows first, to factor x what that cell needs (exact: no arrival is ever dropped
 *     for want of slots below 65280 per cell);
 *   * pmcx_timestep, pmcx_boundary_resample: at the start of the call, (i) when the fullest cell uses
 *     more than fill x cap slots, to factor x that, and (ii) when some cell's slots in use plus the sum
 *     of its classes' ideal counts (pmcx_set_n_part_ideal: what a doubling, or the weight preparation
 *     of an add, can bring a cell in one call) exceed cap, to 1.25 x that -- room for the step's
 *     refill and doubling: a margin, not a count;
 *   * pmcx_scenario_update_aero_state, pmcx_seasalt_emissions (also inside pmcx_timestep): before the
 *     adds, an upper bound of what they can bring a cell to -- per mode the Poisson mean of its sample
 *     + 6 sigma + 12, with the weight preparation on also the classes' ideal counts -- against cap; to
 *     1.25 x that (what still does not fit -- one sample in 10^9 -- raises PMCX_ST_CAP_OVERFLOW as ever);
 *   * pmcx_transport_import: a neighbouring rank's cell larger than a ghost cell here.
 * Each check reads one number back from the device (the call waits for the stream there).  Off by
 * default; the Fortran shim switches it on (fill 0.75, factor 1.5).
 * pmcx_get_growth: the present cap and how many times the block has grown. */
int pmcx_grow(pmcx_ctx *ctx, int32_t new_cap);
int pmcx_set_auto_grow(pmcx_ctx *ctx, int32_t on, double fill, double factor);
int pmcx_get_growth(pmcx_ctx *ctx, int32_t *cap, int32_t *n_grown);

/* Global id of local cell 0 (the RNG key, pmcx_config.cell_id_base) of an existing context:
 * what a per-cell caller changes between cells instead of rebuilding the context. */
int pmcx_set_cell_id_base(pmcx_ctx *ctx, uint32_t cell_id_base);

/* One partmc_timestep loop body over all cells (wrf_pmc_driver.F90:169-254), in the
 * reference's order:
 *   mc_coag when do_coagulation and mod(elapsed_time, partmc_chem_dt) == 0     driver:189-198
 *   scenario_update_gas_state (when a gas scenario is resident and height given) driver:202-205
 *   scenario_update_aero_state (when any of the three aerosol inputs is given)  driver:207-210
 *   seasalt_emissions on the ocean surface cells (when sea_salt_opt > 0)         driver:212-223
 *   aero_state_rebalance                                                        driver:248-250
 * over the cells of pmcx_set_loop_cells.  MOSAIC (driver:232-245) is outside this library.  RNG
 * step index: chemistry-step count for mc_coag, WRF-step count (elapsed_time / dt) for the rest. */
typedef struct {
    int32_t struct_size;     /* sizeof(pmcx_step_flags): checked (pmcx_step_flags_default sets it) */
    int32_t do_coagulation;  /* grid%do_coagulation                       */
    int32_t allow_doubling;  /* grid%allow_doubling                       */
    int32_t allow_halving;   /* grid%allow_halving                        */
    int32_t kernel_type;     /* COAG_KERNEL_TYPE_BROWN in the reference   */
    double dt;               /* grid%dt (WRF step)                        */
    double partmc_chem_dt;   /* grid%partmc_chem_dt                       */
    double elapsed_time;     /* env_state%elapsed_time at this call       */
    uint64_t random_seed;    /* grid%random_seed                          */
    /* scenario(i,k,j) at this step; NULL = nothing of that kind */
    const double *height;                  /* env_state%height [n_cells]: enables the gas update */
    const double *old_height;              /* old_env_state%height [n_cells] or NULL             */
    const pmcx_aero_dist *aero_emission;   /* as pmcx_scenario_update_aero_state                 */
    const double *aero_dilution_rate;      /* [n_cells]                                          */
    const pmcx_aero_dist *aero_background;
    /* seasalt_emissions between the scenario update and the rebalance (driver:212-223), when
     * sea_salt_opt = grid%partmc_seasalt_param > 0: arguments as pmcx_seasalt_emissions (height above
     * is env_state%height and is required then) */
    int32_t sea_salt_opt, i_na, i_cl, pad_;
    const int32_t *ocean_surface;          /* [n_cells] k == 1 .and. xland > 1.5                 */
    const double *wind_speed;              /* [n_cells]                                           */
    const int32_t *ss_mode_class, *ss_mode_source;
} pmcx_step_flags;

/* zeros, struct_size, kernel_type = COAG_KERNEL_TYPE_BROWN, do_coagulation = allow_doubling =
 * allow_halving = 1 (registry.partmc:2,14-15), partmc_chem_dt = 60 (registry.partmc:24) * The step has ONE status word per cell: it is started here and every stage adds its bits (called by
 * itself each stage starts its own), so that what mc_coag reported -- a full event log, a cell out of
 * slots -- is still there when the step returns (pmcx_get_coag_counters).
 */
void pmcx_step_flags_default(pmcx_step_flags *flags);

int pmcx_timestep(pmcx_ctx *ctx, const pmcx_step_flags *flags);

/* ---- introspection used by tests and by the roofline accounting ---- */

int pmcx_get_bin_tables(pmcx_ctx *ctx, double *v_edge /*[n_bin+1]*/,
                        double *v_samp /*[n_bin*3]*/);
int pmcx_get_diag_grid(pmcx_ctx *ctx, double *d_edge /*[n_dbin+1]*/,
                       double *d_width /*[n_dbin]*/);
int pmcx_get_kmax(pmcx_ctx *ctx, int32_t cell, double *kmax /*[n_bin*n_bin]*/);
/* cached per-particle totals: vtot, mass [cap] of one cell */
int pmcx_get_caches(pmcx_ctx *ctx, int32_t cell, double *vtot, double *mass);
/* aero_sorted of one cell: the slot lists perm[pcap], pcap = min(65535, cap + cap/8 +
 * min(16 G, 8192)), G = n_class n_bin; list lengths gcnt[G]; list starts goff[G + 1].  Each list has
 * free entries behind it -- 16 + n/8, behind the empty groups 16, while perm has the room; with tens of
 * weight classes or a cell near 65 280 slots the room shrinks, first behind the empty groups, by a
 * fixed ladder (16 + n/8 | 4, | 1, | 0, 4 + n/8 | 0, n/8 | 0, none): a merged particle that finds no
 * room in its new group's list goes to the table of appends (max_appends), as ever.  The lists
 * persist between mc_coag calls like aero_state%valid_sort
 * (wrf_pmc_trans_aero.F90:2094): pmcx_sort_only sorts only the cells whose
 * population was changed by anything other than mc_coag since their last sort. */
int pmcx_sort_only(pmcx_ctx *ctx);
/* aero_state%valid_sort = .false. for every cell (what a host-side edit of the
 * particles implies; also used to time the sort on every step) */
int pmcx_invalidate_sort(pmcx_ctx *ctx);
/* Diagnostic: pin the pair sampler's launch shape instead of the library's own choice by
 * domain size, so that tests can hold every variant to the oracle at small sizes.  lanes: 0
 * automatic (with several weight classes: 5); 1 lane per cell (the full domain's kernel: with one
 * class and the Brownian kernel its draw runs ahead of the sweep as a phase of its own); 3 lane per cell
 * without that; 2 its two-wavefront form (a rank's share at N = 4, 8); 4 lane per cell with the 256 cells
 * of a workgroup grouped every round by what they do next (k_coag_grp; one class, Brownian);
 * 5 several classes: a wavefront per cell screens the slots of all class pairs, a lane per cell sweeps
 * the list of those that may draw samples, in rounds until no cell is left (one host synchronisation per
 * mc_coag); 8, 16, 64 lanes per cell (small blocks).  A form that does not apply to the configuration
 * falls back to the library's choice.  No form touches the species volumes: they are summed behind the
 * sampler from the events' log (any species count runs the same forms).  wavefronts_per_group: 0
 * automatic, 1, 2 or 4 (the 8/16/64 kernels; with lanes = 1: 1, 2 cap the wavefronts a SIMD
 * holds; with lanes = 5: 1, 2 cut a cell's list to 64, 256 entries, i.e. many rounds).  stamps: 1 = the instantiation that records
 * cycles per stage (pmcx_profile_phase_cycles).  Results do not depend on any of them. */
int pmcx_set_sampler_variant(pmcx_ctx *ctx, int32_t lanes, int32_t wavefronts_per_group, int32_t stamps);

/* Inspection: out[0] = cells the lane-per-cell sampler's split-off draw left to its second launch in
 * the last pmcx_mc_coag (-1: that form did not run); out[1] = the form it ran in (pmcx_set_sampler_variant's
 * numbering: 5 with several weight classes); out[2] = with form 5 the rounds it launched (the last one
 * found no cell left); out[3] reserved (0).  Synchronises. */
int pmcx_get_sampler_info(pmcx_ctx *ctx, int32_t out[4]);

/* The lists only: the library keeps, per slot, the sort key (group, or "tombstone") it was
 * last binned under, and a kernel that changes a population while keeping those keys current
 * leaves exactly this behind -- lists to be rebuilt from keys that hold.  The next sort then
 * streams 2 bytes per particle instead of the particle records (also used to time that). */
int pmcx_invalidate_lists(pmcx_ctx *ctx);
/* inspection (tests): the keys of one cell [cap], whether they hold, the pending event log of
 * the pair sampler [256] (slot | key << 16) and its length; any pointer may be NULL */
int pmcx_get_keys(pmcx_ctx *ctx, int32_t cell, uint16_t *gkey, int32_t *key_valid, uint32_t *evlog,
                  int32_t *n_log);
int pmcx_get_sort(pmcx_ctx *ctx, int32_t cell, uint32_t *perm, int32_t *gcnt,
                  int32_t *goff);

/* Evaluate a device math routine elementwise (bitwise parity tests):
 * op 0 exp, 1 log, 2 cbrt, 3 norminv, 4 sqrt, 5 a/b (in2 = b),
 * 6 poisson(mean=in, u=in2), 7 kernel K(v1=in, m1=in2, v2=in3, m2=in4)
 * at (temp, pressure) with kernel id ktype, 8 dry-deposition velocity
 * (diameter=in, density=in2, ustar=in3, aer_res_a=in4; pressure carries
 * 1/rho_air; ktype 0: alpha 1.2, A 0.002; 1: alpha 100, A 9.999), 9 the single-precision
 * screen of the Brownian kernel the lane-per-cell sampler consults before the fp64 value
 * (same operands as 7; -1 where the screen declines to answer). */
int pmcx_test_math(pmcx_ctx *ctx, int32_t op, int64_t n, const double *in,
                   const double *in2, const double *in3, const double *in4,
                   int32_t ktype, double temp, double pressure, double *out);
int pmcx_test_philox(pmcx_ctx *ctx, const uint32_t ctr[4], const uint32_t key[2],
                     uint32_t out[4]);

/* Per-kernel device timing with HIP events on the context's stream.
 * names: "bin_sort", "coag_sample", "compact", "rebalance", "diagnostics",
 * "kernel_bounds", "generate".  Returns the accumulated milliseconds and
 * launch count since the last reset for one name. */
int pmcx_profile_enable(pmcx_ctx *ctx, int32_t on);
int pmcx_profile_reset(pmcx_ctx *ctx);
int pmcx_profile_get(pmcx_ctx *ctx, const char *name, double *ms, int64_t *launches);

/* Diagnostic: s_memtime ticks per stage of the lane-per-cell sampler's pass, summed over
 * wavefronts, after pmcx_set_sampler_variant(ctx, …, 1) (a separate, slower
 * instantiation of the kernel; zeros otherwise).  out[0] set-up; out[1..7] the stages of a pass
 * (slots settled and blocks fetched; exact counts; pair picked and list entries requested;
 * particle records requested; kernel and accept; unused; coagulation); out[8] write-back;
 * out[9] draws; out[10] passes; out[11] wavefronts. */
int pmcx_profile_phase_cycles(pmcx_ctx *ctx, uint64_t out[12], int32_t reset);

/* device memory held by the context, bytes (replaces memtrack, interface/mem.c) */
int pmcx_memory_bytes(pmcx_ctx *ctx, int64_t *bytes);
/* free and total memory of a HIP device, bytes (no context needed: sizes a block of cells
 * before pmcx_create; 0 on success, PMCX_EDEVICE otherwise) */
int pmcx_device_memory_info(int32_t device, int64_t *free_bytes, int64_t *total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PMCX_H */
