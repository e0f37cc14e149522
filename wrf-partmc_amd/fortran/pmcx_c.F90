! pmcx_c.F90 -- ISO_C_BINDING interfaces of libpmcx.so, one per function of include/pmcx.h, and the
! bind(c) mirrors of its structs.  Nothing else: the procedures a WRF-PartMC maintainer calls are
! in pmcx_shim.F90.
!
! Optional array arguments of the C-ABI (a NULL pointer means "not given") are type(c_ptr) by value
! here -- pass c_loc(array) or c_null_ptr; mandatory ones are assumed-size arrays.
module pmcx_c
  use iso_c_binding
  implicit none
  public

  !> pmcx_config (include/pmcx.h)
  type, bind(c) :: pmcx_config
     integer(c_int32_t) :: struct_size, device, n_cells, cap, n_spec, n_class, n_source, n_bin
     real(c_double) :: r_min, r_max
     integer(c_int32_t) :: max_events, max_appends, n_dbin, i_water
     real(c_double) :: d_min, d_max
     integer(c_int32_t) :: cell_id_base, compact_den
     real(c_double) :: const_k, additive_beta
     integer(c_int32_t) :: n_comp_max, pad_
  end type pmcx_config

  !> pmcx_trans_fields: WRF's fields for the transport probabilities
  type, bind(c) :: pmcx_trans_fields_t
     integer(c_int32_t) :: nx, nz, ny, do_vertmix, all_columns, i_offset, j_offset, global_nx, global_ny
     real(c_double) :: dt, dx, dy
     type(c_ptr) :: alt, xkhh, u_flux, v_flux, w_flux, z, z_at_w, exch_h
  end type pmcx_trans_fields_t

  !> pmcx_aero_dist: log-normal modes with a number concentration per cell and mode
  type, bind(c) :: pmcx_aero_dist_t
     integer(c_int32_t) :: n_mode
     type(c_ptr) :: num_conc, gm_radius, gsd_log10, vol_frac, mode_class, mode_source
     integer(c_int32_t) :: per_cell
  end type pmcx_aero_dist_t

  !> pmcx_step_flags: what one partmc_timestep loop body needs
  type, bind(c) :: pmcx_step_flags_t
     integer(c_int32_t) :: struct_size, do_coagulation, allow_doubling, allow_halving, kernel_type
     real(c_double) :: dt, partmc_chem_dt, elapsed_time
     integer(c_int64_t) :: random_seed
     type(c_ptr) :: height, old_height, aero_emission, aero_dilution_rate, aero_background
     integer(c_int32_t) :: sea_salt_opt, i_na, i_cl, pad_
     type(c_ptr) :: ocean_surface, wind_speed, ss_mode_class, ss_mode_source
  end type pmcx_step_flags_t

  interface
     ! ---- context -------------------------------------------------------------------------
     subroutine pmcx_config_default(cfg) bind(c, name="pmcx_config_default")
       import :: pmcx_config
       type(pmcx_config), intent(out) :: cfg
     end subroutine pmcx_config_default
     integer(c_int) function pmcx_create(cfg, ctx) bind(c, name="pmcx_create")
       import :: pmcx_config, c_ptr, c_int
       type(pmcx_config), intent(in) :: cfg
       type(c_ptr), intent(out) :: ctx
     end function pmcx_create
     integer(c_int) function pmcx_destroy(ctx) bind(c, name="pmcx_destroy")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_destroy
     function pmcx_last_error(ctx) bind(c, name="pmcx_last_error") result(msg)
       import :: c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr) :: msg
     end function pmcx_last_error
     integer(c_int) function pmcx_synchronize(ctx) bind(c, name="pmcx_synchronize")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_synchronize
     integer(c_int32_t) function pmcx_max_cap(cfg) bind(c, name="pmcx_max_cap")
       import :: pmcx_config, c_int32_t
       type(pmcx_config), intent(in) :: cfg
     end function pmcx_max_cap
     integer(c_int) function pmcx_grow(ctx, new_cap) bind(c, name="pmcx_grow")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: new_cap
     end function pmcx_grow
     integer(c_int) function pmcx_set_auto_grow(ctx, on, fill, factor) bind(c, name="pmcx_set_auto_grow")
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
       real(c_double), value :: fill, factor
     end function pmcx_set_auto_grow
     integer(c_int) function pmcx_get_growth(ctx, cap, n_grown) bind(c, name="pmcx_get_growth")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: cap, n_grown
     end function pmcx_get_growth
     integer(c_int) function pmcx_memory_bytes(ctx, bytes) bind(c, name="pmcx_memory_bytes")
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: bytes
     end function pmcx_memory_bytes
     integer(c_int) function pmcx_device_memory_info(device, free_bytes, total_bytes) &
          bind(c, name="pmcx_device_memory_info")
       import :: c_int, c_int32_t, c_int64_t
       integer(c_int32_t), value :: device
       integer(c_int64_t), intent(out) :: free_bytes, total_bytes
     end function pmcx_device_memory_info

     ! ---- model, environment, weights -----------------------------------------------------
     integer(c_int) function pmcx_set_aero_data(ctx, density, kappa, molec_weight) &
          bind(c, name="pmcx_set_aero_data")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: density(*)
       type(c_ptr), value :: kappa, molec_weight
     end function pmcx_set_aero_data
     integer(c_int) function pmcx_set_env(ctx, temp, pressure, rel_humid) bind(c, name="pmcx_set_env")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: temp(*), pressure(*), rel_humid(*)
     end function pmcx_set_env
     integer(c_int) function pmcx_set_weights(ctx, magnitude) bind(c, name="pmcx_set_weights")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: magnitude(*)
     end function pmcx_set_weights
     integer(c_int) function pmcx_get_weights(ctx, magnitude) bind(c, name="pmcx_get_weights")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: magnitude(*)
     end function pmcx_get_weights
     integer(c_int) function pmcx_set_n_part_ideal(ctx, n_ideal) bind(c, name="pmcx_set_n_part_ideal")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: n_ideal(*)
     end function pmcx_set_n_part_ideal
     integer(c_int) function pmcx_set_weight_preparation(ctx, on, allow_doubling, allow_halving) &
          bind(c, name="pmcx_set_weight_preparation")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on, allow_doubling, allow_halving
     end function pmcx_set_weight_preparation
     integer(c_int) function pmcx_set_transport_law(ctx, law) bind(c, name="pmcx_set_transport_law")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: law
     end function pmcx_set_transport_law
     integer(c_int) function pmcx_set_cell_id_base(ctx, cell_id_base) bind(c, name="pmcx_set_cell_id_base")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cell_id_base
     end function pmcx_set_cell_id_base
     integer(c_int) function pmcx_set_next_id(ctx, next_id) bind(c, name="pmcx_set_next_id")
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(in) :: next_id(*)
     end function pmcx_set_next_id
     integer(c_int) function pmcx_get_next_id(ctx, next_id) bind(c, name="pmcx_get_next_id")
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: next_id(*)
     end function pmcx_get_next_id
     integer(c_int) function pmcx_set_loop_cells(ctx, in_loop) bind(c, name="pmcx_set_loop_cells")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, in_loop       ! int32 [n_cells] or c_null_ptr
     end function pmcx_set_loop_cells

     ! ---- particles -----------------------------------------------------------------------
     integer(c_int) function pmcx_upload_cells(ctx, cell_first, n_up, part_offset, vol, wclass, &
          id, least_ct, greatest_ct, n_orig) bind(c, name="pmcx_upload_cells")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cell_first, n_up
       integer(c_int64_t), intent(in) :: part_offset(*), id(*)
       real(c_double), intent(in) :: vol(*), least_ct(*), greatest_ct(*)
       integer(c_int32_t), intent(in) :: wclass(*)
       type(c_ptr), value :: n_orig
     end function pmcx_upload_cells
     integer(c_int) function pmcx_get_n_part(ctx, n_part) bind(c, name="pmcx_get_n_part")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: n_part(*)
     end function pmcx_get_n_part
     integer(c_int) function pmcx_download_cells(ctx, cell_first, n_dn, part_offset, vol, wclass, &
          id, least_ct, greatest_ct, n_orig) bind(c, name="pmcx_download_cells")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cell_first, n_dn
       integer(c_int64_t), intent(in) :: part_offset(*)
       integer(c_int64_t), intent(out) :: id(*)
       real(c_double), intent(out) :: vol(*), least_ct(*), greatest_ct(*)
       integer(c_int32_t), intent(out) :: wclass(*)
       type(c_ptr), value :: n_orig
     end function pmcx_download_cells
     integer(c_int) function pmcx_upload_components(ctx, cell_first, n_up, part_offset, comp_len, &
          comp_source, comp_time) bind(c, name="pmcx_upload_components")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cell_first, n_up
       integer(c_int64_t), intent(in) :: part_offset(*)
       integer(c_int32_t), intent(in) :: comp_len(*), comp_source(*)
       real(c_double), intent(in) :: comp_time(*)
     end function pmcx_upload_components
     integer(c_int) function pmcx_download_components(ctx, cell_first, n_dn, part_offset, comp_len, &
          comp_source, comp_time) bind(c, name="pmcx_download_components")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cell_first, n_dn
       integer(c_int64_t), intent(in) :: part_offset(*)
       integer(c_int32_t), intent(out) :: comp_len(*), comp_source(*)
       real(c_double), intent(out) :: comp_time(*)
     end function pmcx_download_components
     integer(c_int) function pmcx_generate_lognormal(ctx, n_part, n_mode, mode_frac, gm_radius, &
          gsd_log10, vol_frac, mode_class, mode_source, seed, create_time) &
          bind(c, name="pmcx_generate_lognormal")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n_part, n_mode
       real(c_double), intent(in) :: mode_frac(*), gm_radius(*), gsd_log10(*), vol_frac(*)
       type(c_ptr), value :: mode_class, mode_source
       integer(c_int64_t), value :: seed
       real(c_double), value :: create_time
     end function pmcx_generate_lognormal

     ! ---- mc_coag -------------------------------------------------------------------------
     integer(c_int) function pmcx_update_kernel_bounds(ctx, kernel_type) &
          bind(c, name="pmcx_update_kernel_bounds")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: kernel_type
     end function pmcx_update_kernel_bounds
     integer(c_int) function pmcx_set_kernel_bounds_margin(ctx, margin) &
          bind(c, name="pmcx_set_kernel_bounds_margin")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: margin
     end function pmcx_set_kernel_bounds_margin
     integer(c_int) function pmcx_mc_coag(ctx, kernel_type, del_t, seed, step) bind(c, name="pmcx_mc_coag")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: kernel_type
       real(c_double), value :: del_t
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
     end function pmcx_mc_coag
     integer(c_int) function pmcx_get_coag_counters(ctx, n_samp, n_coag, status) &
          bind(c, name="pmcx_get_coag_counters")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: n_samp(*), n_coag(*), status(*)
     end function pmcx_get_coag_counters
     integer(c_int) function pmcx_get_coag_totals(ctx, n_samp, n_coag, n_part, status_or) &
          bind(c, name="pmcx_get_coag_totals")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: n_samp, n_coag, n_part
       integer(c_int32_t), intent(out) :: status_or
     end function pmcx_get_coag_totals
     integer(c_int) function pmcx_accumulate_totals(ctx) bind(c, name="pmcx_accumulate_totals")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_accumulate_totals
     integer(c_int) function pmcx_read_accumulated(ctx, n_samp, n_coag, n_part, status_or, reset) &
          bind(c, name="pmcx_read_accumulated")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: n_samp, n_coag, n_part
       integer(c_int32_t), intent(out) :: status_or
       integer(c_int32_t), value :: reset
     end function pmcx_read_accumulated
     integer(c_int) function pmcx_sort_only(ctx) bind(c, name="pmcx_sort_only")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_sort_only
     integer(c_int) function pmcx_invalidate_sort(ctx) bind(c, name="pmcx_invalidate_sort")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_invalidate_sort

     ! ---- the rest of the loop body -------------------------------------------------------
     integer(c_int) function pmcx_rebalance(ctx, allow_doubling, allow_halving, seed, step) &
          bind(c, name="pmcx_rebalance")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: allow_doubling, allow_halving
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
     end function pmcx_rebalance
     integer(c_int) function pmcx_scenario_update_aero_state(ctx, dt, elapsed_time, seed, step, &
          emission, dilution_rate, background, n_emit, n_dil_in, n_dil_out, height, old_height) &
          bind(c, name="pmcx_scenario_update_aero_state")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt, elapsed_time
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       type(c_ptr), value :: emission, dilution_rate, background   ! pmcx_aero_dist_t*, double*, ...
       type(c_ptr), value :: n_emit, n_dil_in, n_dil_out           ! int32 [n_cells] or c_null_ptr
       type(c_ptr), value :: height, old_height                    ! double [n_cells] or c_null_ptr
     end function pmcx_scenario_update_aero_state
     integer(c_int) function pmcx_set_aero_scenario_modes(ctx, which, dist) &
          bind(c, name="pmcx_set_aero_scenario_modes")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: which
       type(c_ptr), value :: dist                  ! pmcx_aero_dist_t* or c_null_ptr
     end function pmcx_set_aero_scenario_modes
     integer(c_int) function pmcx_seasalt_emissions(ctx, dt, elapsed_time, seed, step, sea_salt_opt, &
          i_na, i_cl, ocean_surface, wind_speed, height, mode_class, mode_source, n_emit, rate) &
          bind(c, name="pmcx_seasalt_emissions")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt, elapsed_time
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step, sea_salt_opt, i_na, i_cl
       integer(c_int32_t), intent(in) :: ocean_surface(*)
       real(c_double), intent(in) :: wind_speed(*), height(*)
       integer(c_int32_t), intent(in) :: mode_class(*), mode_source(*)
       integer(c_int32_t), intent(out) :: n_emit(*)
       type(c_ptr), value :: rate
     end function pmcx_seasalt_emissions
     integer(c_int) function pmcx_set_gas_state(ctx, n_gas, mix_rat) bind(c, name="pmcx_set_gas_state")
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n_gas
       real(c_double), intent(in) :: mix_rat(*)
     end function pmcx_set_gas_state
     integer(c_int) function pmcx_get_gas_state(ctx, mix_rat) bind(c, name="pmcx_get_gas_state")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: mix_rat(*)
     end function pmcx_get_gas_state
     integer(c_int) function pmcx_set_gas_scenario(ctx, n_time_e, emission_time, emission_rate_scale, &
          emission, n_time_d, dilution_time, dilution_rate, background) &
          bind(c, name="pmcx_set_gas_scenario")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n_time_e, n_time_d
       type(c_ptr), value :: emission_time, emission_rate_scale, emission, dilution_time, &
            dilution_rate, background
     end function pmcx_set_gas_scenario
     integer(c_int) function pmcx_scenario_update_gas_state(ctx, dt, elapsed_time, height, old_height) &
          bind(c, name="pmcx_scenario_update_gas_state")
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt, elapsed_time
       real(c_double), intent(in) :: height(*)
       type(c_ptr), value :: old_height
     end function pmcx_scenario_update_gas_state
     subroutine pmcx_step_flags_default(flags) bind(c, name="pmcx_step_flags_default")
       import :: pmcx_step_flags_t
       type(pmcx_step_flags_t), intent(out) :: flags
     end subroutine pmcx_step_flags_default
     integer(c_int) function pmcx_timestep(ctx, flags) bind(c, name="pmcx_timestep")
       import :: c_ptr, c_int, pmcx_step_flags_t
       type(c_ptr), value :: ctx
       type(pmcx_step_flags_t), intent(in) :: flags
     end function pmcx_timestep
     integer(c_int) function pmcx_condense(ctx, del_t, n_sub) bind(c, name="pmcx_condense")
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: del_t
       integer(c_int32_t), value :: n_sub
     end function pmcx_condense

     ! ---- transport driver ----------------------------------------------------------------
     integer(c_int) function pmcx_transport_probs(ctx, fields, prob_h, prob_v, n_substeps, bad, &
          n_class_part) bind(c, name="pmcx_transport_probs")
       import :: c_ptr, c_int, pmcx_trans_fields_t
       type(c_ptr), value :: ctx
       type(pmcx_trans_fields_t), intent(in) :: fields
       type(c_ptr), value :: prob_h, prob_v, n_substeps, bad, n_class_part
     end function pmcx_transport_probs
     integer(c_int) function pmcx_set_transport_lateral(ctx, on, j_global0, global_ny) &
          bind(c, name="pmcx_set_transport_lateral")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on, j_global0, global_ny
     end function pmcx_set_transport_lateral
     integer(c_int) function pmcx_transport_preweight(ctx, nx, nz, ny, periodic_x, periodic_y, prob_h, &
          prob_v, cell_volume, seed, step, expected_num_conc, n_removed) &
          bind(c, name="pmcx_transport_preweight")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, nz, ny, periodic_x, periodic_y
       type(c_ptr), value :: prob_h, prob_v        ! c_null_ptr: the ones pmcx_transport_probs built
       real(c_double), intent(in) :: cell_volume(*)
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       real(c_double), intent(out) :: expected_num_conc(*)
       integer(c_int32_t), intent(out) :: n_removed(*)
     end function pmcx_transport_preweight
     integer(c_int) function pmcx_transport_preweight_begin(ctx, nx, nz, ny, periodic_x, periodic_y, &
          j_global0, global_ny, prob_h, prob_v, cell_volume, edge_terms) &
          bind(c, name="pmcx_transport_preweight_begin")
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, nz, ny, periodic_x, periodic_y, j_global0, global_ny
       type(c_ptr), value :: prob_h, prob_v
       real(c_double), intent(in) :: cell_volume(*)
       real(c_double), intent(out) :: edge_terms(*)
     end function pmcx_transport_preweight_begin
     integer(c_int) function pmcx_transport_preweight_finish(ctx, terms_from_below, terms_from_above, &
          below_is_edge, above_is_edge, seed, step, expected_num_conc, n_removed) &
          bind(c, name="pmcx_transport_preweight_finish")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx, terms_from_below, terms_from_above
       integer(c_int32_t), value :: below_is_edge, above_is_edge
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       real(c_double), intent(out) :: expected_num_conc(*)
       integer(c_int32_t), intent(out) :: n_removed(*)
     end function pmcx_transport_preweight_finish
     integer(c_int) function pmcx_transport(ctx, nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v, &
          cell_volume, seed, step, n_moved, n_arrived) bind(c, name="pmcx_transport")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, nz, ny, periodic_x, periodic_y
       type(c_ptr), value :: prob_h, prob_v        ! c_null_ptr twice: the probabilities built on the device
       real(c_double), intent(in) :: cell_volume(*)
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       integer(c_int32_t), intent(out) :: n_moved(*), n_arrived(*)
     end function pmcx_transport
     integer(c_int) function pmcx_transport_begin(ctx, nx, nz, ny, periodic_x, prob_h, prob_v, &
          cell_volume, seed, step) bind(c, name="pmcx_transport_begin")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, nz, ny, periodic_x
       type(c_ptr), value :: prob_h, prob_v
       real(c_double), intent(in) :: cell_volume(*)
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
     end function pmcx_transport_begin
     integer(c_int) function pmcx_transport_export_size(ctx, side, n_bytes) &
          bind(c, name="pmcx_transport_export_size")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: side
       integer(c_int64_t), intent(out) :: n_bytes
     end function pmcx_transport_export_size
     integer(c_int) function pmcx_transport_export(ctx, side, dev_buf) bind(c, name="pmcx_transport_export")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx, dev_buf
       integer(c_int32_t), value :: side
     end function pmcx_transport_export
     integer(c_int) function pmcx_transport_import(ctx, side, dev_buf, n_bytes) &
          bind(c, name="pmcx_transport_import")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx, dev_buf
       integer(c_int32_t), value :: side
       integer(c_int64_t), value :: n_bytes
     end function pmcx_transport_import
     integer(c_int) function pmcx_transport_finish(ctx, n_moved, n_arrived) &
          bind(c, name="pmcx_transport_finish")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: n_moved(*), n_arrived(*)
     end function pmcx_transport_finish
     integer(c_int) function pmcx_boundary_copy(ctx, n_pairs, from_cell, to_cell) &
          bind(c, name="pmcx_boundary_copy")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n_pairs
       integer(c_int32_t), intent(in) :: from_cell(*), to_cell(*)
     end function pmcx_boundary_copy
     integer(c_int) function pmcx_boundary_resample(ctx, n_list, cell, throw_away, background, density, &
          create_time, seed, step, n_added, n_removed) bind(c, name="pmcx_boundary_resample")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double, pmcx_aero_dist_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n_list
       integer(c_int32_t), intent(in) :: cell(*), throw_away(*)
       type(pmcx_aero_dist_t), intent(in) :: background
       real(c_double), intent(in) :: density(*)
       real(c_double), value :: create_time
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       integer(c_int32_t), intent(out) :: n_added(*), n_removed(*)
     end function pmcx_boundary_resample
     integer(c_int) function pmcx_dry_deposition(ctx, dt, seed, step, surface, aer_res_a, ustar, alpha, &
          big_a, height, inv_density, n_removed, unrealistic) bind(c, name="pmcx_dry_deposition")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
       integer(c_int32_t), intent(in) :: surface(*)
       real(c_double), intent(in) :: aer_res_a(*), ustar(*), alpha(*), big_a(*), height(*), inv_density(*)
       integer(c_int32_t), intent(out) :: n_removed(*), unrealistic(*)
     end function pmcx_dry_deposition
     integer(c_int) function pmcx_device_alloc(ctx, n_bytes, dev_ptr) bind(c, name="pmcx_device_alloc")
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: n_bytes
       type(c_ptr), intent(out) :: dev_ptr
     end function pmcx_device_alloc
     integer(c_int) function pmcx_device_free(ctx, dev_ptr) bind(c, name="pmcx_device_free")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, dev_ptr
     end function pmcx_device_free
     integer(c_int) function pmcx_set_transport_cuts(ctx, cut_x, nx_local, i_global0, global_nx) &
          bind(c, name="pmcx_set_transport_cuts")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: cut_x, nx_local, i_global0, global_nx
     end function pmcx_set_transport_cuts
     integer(c_int) function pmcx_transport_preweight_x_terms(ctx, terms_from_west, terms_from_east, &
          west_is_edge, east_is_edge) bind(c, name="pmcx_transport_preweight_x_terms")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx, terms_from_west, terms_from_east
       integer(c_int32_t), value :: west_is_edge, east_is_edge
     end function pmcx_transport_preweight_x_terms
     integer(c_int) function pmcx_set_async(ctx, on) bind(c, name="pmcx_set_async")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function pmcx_set_async
     integer(c_int) function pmcx_host_alloc(ctx, n_bytes, host_ptr) bind(c, name="pmcx_host_alloc")
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: n_bytes
       type(c_ptr), intent(out) :: host_ptr
     end function pmcx_host_alloc
     integer(c_int) function pmcx_host_free(ctx, host_ptr) bind(c, name="pmcx_host_free")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, host_ptr
     end function pmcx_host_free
     integer(c_int) function pmcx_device_copy(ctx, dst, src, n_bytes, kind) bind(c, name="pmcx_device_copy")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx, dst, src
       integer(c_int64_t), value :: n_bytes
       integer(c_int32_t), value :: kind
     end function pmcx_device_copy

     ! ---- partmc_process observables ------------------------------------------------------
     integer(c_int) function pmcx_diagnostics(ctx, tot_num_conc, tot_mass_conc, num_dist, mass_dist) &
          bind(c, name="pmcx_diagnostics")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, tot_num_conc, tot_mass_conc, num_dist, mass_dist
     end function pmcx_diagnostics
     integer(c_int) function pmcx_diagnostics_ext(ctx, species_mass_conc, num_conc_by_source, &
          pm_mass_conc, n_parts) bind(c, name="pmcx_diagnostics_ext")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, species_mass_conc, num_conc_by_source, pm_mass_conc, n_parts
     end function pmcx_diagnostics_ext

     ! ---- timing buckets (MPI_Wtime pairs of wrf_pmc_driver.F90:148-153,191-196) -----------
     integer(c_int) function pmcx_profile_enable(ctx, on) bind(c, name="pmcx_profile_enable")
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function pmcx_profile_enable
     integer(c_int) function pmcx_profile_reset(ctx) bind(c, name="pmcx_profile_reset")
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function pmcx_profile_reset
     integer(c_int) function pmcx_profile_get(ctx, name, ms, launches) bind(c, name="pmcx_profile_get")
       import :: c_ptr, c_int, c_char, c_double, c_int64_t
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)
       real(c_double), intent(out) :: ms
       integer(c_int64_t), intent(out) :: launches
     end function pmcx_profile_get
  end interface

end module pmcx_c
