! pmcx_shim.F90 -- ISO_C_BINDING shim between the WRF-PartMC Fortran host and
! libpmcx.so (include/pmcx.h).
!
! Keeps the call surface of the reference's per-cell loop body
! (interface/wrf_pmc_driver.F90:169-254):
!
!   call mc_coag(COAG_KERNEL_TYPE_BROWN, env_states(i,k,j), aero_data, &
!        aero_states(i,k,j), chemistry_call, n_samp, n_coag)      driver:192-194
!
! becomes, per cell (drop-in, same argument list)
!
!   call mc_coag_pmcx(COAG_KERNEL_TYPE_BROWN, env_states(i,k,j), aero_data, &
!        aero_states(i,k,j), chemistry_call, n_samp, n_coag)
!
! or, batched over every owned cell in one call (the fast path: one upload,
! one launch set, one download for the whole rank)
!
!   call partmc_coag_cells_pmcx(COAG_KERNEL_TYPE_BROWN, env_states, aero_data, &
!        aero_states, chemistry_call, n_samp(:), n_coag(:))
!
! aero_state_t holds allocatable components and is not C-interoperable, so
! the shim gathers the particle fields into flat arrays (the layout
! pmcx_upload_cells documents) and scatters them back.  Errors follow the
! reference's abort convention: a non-zero status stops the run (replace
! pmcx_die by wrf_error_fatal when building inside WRF).
!
! `use pmc_mock` is the stand-in for `use pmc_aero_state; use pmc_env_state;
! use pmc_aero_data` when built outside WRF.
!
! What crosses the boundary per particle (wrf_pmc_driver.F90:1419-1466): vol(:), weight_class,
! id, least/greatest_create_time and component(:) -- the latter as a per-source histogram
! (the device keeps counts per source, not the list: DESIGN.md 3.2).  Everything else a
! particle carries on the Fortran side (water_hyst_leg, weight_group, the optics arrays,
! core_vol, the create_time of each constituent) stays on the host and is re-attached BY ID
! after a download; a particle whose composition changed gets its optics reset, a particle the
! host has never seen (emission, transport arrival, doubling) starts from defaults, and
! constituents gained in a coagulation are appended to component(:) with a create_time
! bounded by the particle's least/greatest create time (the exact value is not carried).
module pmcx_shim
  use iso_c_binding
  use pmc_mock
  implicit none
  private
  public :: pmcx_shim_init, pmcx_shim_finalize, pmcx_set_rng, mc_coag_pmcx, &
       partmc_coag_cells_pmcx, aero_state_rebalance_cells_pmcx, aero_state_rebalance_pmcx, &
       pmcx_config
  ! the device-resident sequence (INTEGRATION.md 2c): upload once, step on the device,
  ! download when the host needs the particles (output, restart)
  public :: pmcx_upload_states, pmcx_download_states, mc_coag_resident_pmcx, &
       trans_cells_pmcx, dry_dep_cells_pmcx, trans_preweight_cells_pmcx, &
       trans_from_fields_cells_pmcx, seasalt_cells_pmcx
  ! slab transport across MPI ranks: the host moves the packed buffers (GPU-aware MPI on
  ! the device pointers, or staged through pmcx_device_copy)
  public :: pmcx_shim_context, pmcx_transport_begin, pmcx_transport_export_size, &
       pmcx_transport_export, pmcx_transport_import, pmcx_transport_finish, &
       pmcx_device_alloc, pmcx_device_free, pmcx_device_copy

  type, bind(c) :: pmcx_config
     integer(c_int32_t) :: struct_size, device, n_cells, cap, n_spec, n_class, n_source, n_bin
     real(c_double) :: r_min, r_max
     integer(c_int32_t) :: max_events, max_appends, n_dbin, i_water
     real(c_double) :: d_min, d_max
     integer(c_int32_t) :: cell_id_base, compact_den
     real(c_double) :: const_k, additive_beta
  end type pmcx_config

  !> pmcx_trans_fields (include/pmcx.h): WRF's fields for the transport probabilities
  type, bind(c) :: pmcx_trans_fields_t
     integer(c_int32_t) :: nx, nz, ny, do_vertmix, all_columns, i_offset, j_offset, global_nx, global_ny
     real(c_double) :: dt, dx, dy
     type(c_ptr) :: alt, xkhh, u_flux, v_flux, w_flux, z, z_at_w, exch_h
  end type pmcx_trans_fields_t

  interface
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
     integer(c_int) function pmcx_rebalance(ctx, allow_doubling, allow_halving, seed, step) &
          bind(c, name="pmcx_rebalance")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: allow_doubling, allow_halving
       integer(c_int64_t), value :: seed
       integer(c_int32_t), value :: step
     end function pmcx_rebalance
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
     integer(c_int) function pmcx_transport_probs(ctx, fields, prob_h, prob_v, n_substeps, bad, &
          n_class_part) bind(c, name="pmcx_transport_probs")
       import :: c_ptr, c_int, pmcx_trans_fields_t
       type(c_ptr), value :: ctx
       type(pmcx_trans_fields_t), intent(in) :: fields
       type(c_ptr), value :: prob_h, prob_v, n_substeps, bad, n_class_part
     end function pmcx_transport_probs
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
     integer(c_int) function pmcx_transport_begin(ctx, nx, nz, ny, periodic_x, prob_h, prob_v, &
          cell_volume, seed, step) bind(c, name="pmcx_transport_begin")
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nx, nz, ny, periodic_x
       real(c_double), intent(in) :: prob_h(*), prob_v(*), cell_volume(*)
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
     integer(c_int) function pmcx_device_copy(ctx, dst, src, n_bytes, kind) bind(c, name="pmcx_device_copy")
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx, dst, src
       integer(c_int64_t), value :: n_bytes
       integer(c_int32_t), value :: kind
     end function pmcx_device_copy
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
  end interface

  ! one context per (rank, cell count); rebuilt when the shape changes
  type(c_ptr), save :: ctx = c_null_ptr
  type(pmcx_config), save :: cfg
  logical, save :: have_ctx = .false.
  integer(c_int64_t), save :: rng_seed = 0     ! grid%random_seed (wrf_pmc_init.F90:131)
  integer(c_int32_t), save :: rng_step = 0     ! chemistry-step index
  integer(c_int32_t), save :: cell_id_base = 0 ! global id of this rank's first cell

contains

  subroutine pmcx_die(code, where)
    integer(c_int), intent(in) :: code
    character(len=*), intent(in) :: where
    type(c_ptr) :: cmsg
    character(kind=c_char), pointer :: s(:)
    integer :: n
    if (code == 0) return
    cmsg = pmcx_last_error(ctx)
    write(0,'(a,a,a,i0)') 'pmcx_shim: ', where, ' failed with status ', code
    if (c_associated(cmsg)) then
       call c_f_pointer(cmsg, s, [512])
       n = 1
       do while (n < 512 .and. s(n) /= c_null_char)
          n = n + 1
       end do
       write(0,'(512a1)') s(1:n-1)
    end if
    error stop 1   ! inside WRF: call wrf_error_fatal(...)
  end subroutine pmcx_die

  !> Seed / step of the counter-based RNG and the global id of the first owned
  !> cell (replaces pmc_srand(seed, rank), wrf_pmc_init.F90:131).
  subroutine pmcx_set_rng(seed, step, first_cell_id)
    integer(kind=8), intent(in) :: seed
    integer, intent(in) :: step
    integer, intent(in), optional :: first_cell_id
    rng_seed = seed
    rng_step = step
    if (present(first_cell_id)) then
       cell_id_base = first_cell_id
       ! an existing context is re-keyed in place (no rebuild between the cells of a loop)
       if (have_ctx) then
          cfg%cell_id_base = cell_id_base
          call pmcx_die(pmcx_set_cell_id_base(ctx, cell_id_base), 'pmcx_set_cell_id_base')
       end if
    end if
  end subroutine pmcx_set_rng

  subroutine pmcx_shim_init(n_cells, cap, aero_data, n_class, n_source, device)
    integer, intent(in) :: n_cells, cap, n_class, n_source, device
    type(aero_data_t), intent(in) :: aero_data
    if (have_ctx) call pmcx_shim_finalize()
    call pmcx_config_default(cfg)
    cfg%device = device
    cfg%n_cells = n_cells
    cfg%cap = cap
    cfg%n_spec = size(aero_data%density)
    cfg%n_class = n_class
    cfg%n_source = n_source
    cfg%i_water = aero_data%i_water - 1
    cfg%cell_id_base = cell_id_base
    call pmcx_die(pmcx_create(cfg, ctx), 'pmcx_create')
    have_ctx = .true.
    call pmcx_die(pmcx_set_aero_data(ctx, aero_data%density, c_null_ptr, c_null_ptr), &
         'pmcx_set_aero_data')
  end subroutine pmcx_shim_init

  subroutine pmcx_shim_finalize()
    integer(c_int) :: rc
    if (have_ctx) rc = pmcx_destroy(ctx)
    have_ctx = .false.
    ctx = c_null_ptr
  end subroutine pmcx_shim_finalize

  !> The context handle, for the C entry points the shim has no wrapper for.
  function pmcx_shim_context() result(h)
    type(c_ptr) :: h
    h = ctx
  end function pmcx_shim_context

  !> Gather aero_states(:) into flat arrays (the layout pmcx_upload_cells documents) and
  !> upload them with the cells' environment, weights and ideal particle counts.  cap_min:
  !> room per cell the caller wants beyond the present populations (arrivals of the transport,
  !> emissions); a rebalance may double a cell up to its ideal count, so the capacity is never
  !> below twice that.
  subroutine pmcx_upload_states(env_states, aero_data, aero_states, cap_min)
    type(env_state_t), intent(in) :: env_states(:)
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(in) :: aero_states(:)
    integer, intent(in), optional :: cap_min

    integer :: n_cells, n_spec, n_class, n_source, c, i, s, k, cap_needed
    integer(c_int64_t) :: nt, o
    integer(c_int64_t), allocatable :: part_offset(:), id(:)
    real(c_double), allocatable :: vol(:,:), lct(:), gct(:), temp(:), pres(:), rh(:), mag(:,:), &
         n_ideal(:,:)
    integer(c_int32_t), allocatable, target :: n_orig(:,:)
    integer(c_int32_t), allocatable :: wclass(:)
    type(c_ptr) :: p_norig
    logical :: have_ideal

    n_cells = size(aero_states)
    n_spec = size(aero_data%density)
    n_class = size(aero_states(1)%awa%weight, 2)
    n_source = aero_data_n_source(aero_data)
    cap_needed = 64
    if (present(cap_min)) cap_needed = max(cap_needed, cap_min)
    have_ideal = .true.
    do c = 1, n_cells
       cap_needed = max(cap_needed, aero_states(c)%apa%n_part)
       if (allocated(aero_states(c)%n_part_ideal)) then
          cap_needed = max(cap_needed, 2 * ceiling(sum(aero_states(c)%n_part_ideal(1, :))))
       else
          have_ideal = .false.
       end if
    end do
    cap_needed = min(((cap_needed + 63) / 64) * 64, 65280)
    if (.not. have_ctx .or. cfg%n_cells /= n_cells .or. cfg%cap < cap_needed &
         .or. cfg%n_spec /= n_spec .or. cfg%n_class /= n_class .or. cfg%n_source /= n_source) then
       call pmcx_shim_init(n_cells, cap_needed, aero_data, n_class, n_source, 0)
    end if

    allocate(part_offset(n_cells + 1))
    part_offset(1) = 0
    do c = 1, n_cells
       part_offset(c + 1) = part_offset(c) + aero_states(c)%apa%n_part
    end do
    nt = part_offset(n_cells + 1)
    allocate(vol(max(nt, 1_c_int64_t), n_spec), wclass(max(nt, 1_c_int64_t)), &
         id(max(nt, 1_c_int64_t)), lct(max(nt, 1_c_int64_t)), gct(max(nt, 1_c_int64_t)))
    allocate(n_orig(max(nt, 1_c_int64_t), max(n_source, 1)))
    allocate(temp(n_cells), pres(n_cells), rh(n_cells), mag(n_class, n_cells), &
         n_ideal(n_class, n_cells))
    n_orig = 0
    do c = 1, n_cells
       o = part_offset(c)
       do i = 1, aero_states(c)%apa%n_part
          do s = 1, n_spec
             vol(o + i, s) = aero_states(c)%apa%particle(i)%vol(s)
          end do
          ! component(:) -> constituents per source (driver:1421-1437)
          do k = 1, aero_particle_n_components(aero_states(c)%apa%particle(i))
             s = aero_states(c)%apa%particle(i)%component(k)%source_id
             if (s >= 1 .and. s <= n_source) n_orig(o + i, s) = n_orig(o + i, s) + 1
          end do
          wclass(o + i) = aero_states(c)%apa%particle(i)%weight_class - 1
          id(o + i) = aero_states(c)%apa%particle(i)%id
          lct(o + i) = aero_states(c)%apa%particle(i)%least_create_time
          gct(o + i) = aero_states(c)%apa%particle(i)%greatest_create_time
       end do
       temp(c) = env_states(c)%temp
       pres(c) = env_states(c)%pressure
       rh(c) = env_states(c)%rel_humid
       do s = 1, n_class
          mag(s, c) = aero_states(c)%awa%weight(1, s)%magnitude
          if (have_ideal) n_ideal(s, c) = aero_states(c)%n_part_ideal(1, s)
       end do
    end do
    p_norig = c_null_ptr
    if (n_source > 0) p_norig = c_loc(n_orig)

    call pmcx_die(pmcx_set_env(ctx, temp, pres, rh), 'pmcx_set_env')
    call pmcx_die(pmcx_set_weights(ctx, mag), 'pmcx_set_weights')
    ! aero_state%n_part_ideal (aero_state_set_n_part_ideal, wrf_pmc_init.F90:254): without it
    ! pmcx_rebalance refuses to run (PMCX_ESTATE)
    if (have_ideal) call pmcx_die(pmcx_set_n_part_ideal(ctx, n_ideal), 'pmcx_set_n_part_ideal')
    ! ids of particles created on the device (doubling, emission, transport copies): every cell
    ! draws from its own block of PartMC's counter, which moves past all of them
    block
      integer(c_int64_t), allocatable :: nid(:)
      allocate(nid(n_cells))
      do c = 1, n_cells
         nid(c) = next_id + int(c - 1, 8) * int(2 * cfg%cap, 8)
      end do
      call pmcx_die(pmcx_set_next_id(ctx, nid), 'pmcx_set_next_id')
      next_id = next_id + int(n_cells, 8) * int(2 * cfg%cap, 8)
    end block
    call pmcx_die(pmcx_upload_cells(ctx, 0, n_cells, part_offset, vol, wclass, id, lct, gct, &
         p_norig), 'pmcx_upload_cells')
  end subroutine pmcx_upload_states

  !> ids(order(1)) <= ids(order(2)) <= ... (heap sort of an index array)
  subroutine sort_index_by_id(ids, order)
    integer(kind=8), intent(in) :: ids(:)
    integer, intent(out) :: order(:)
    integer :: n, i, root, child, last, tmp
    n = size(ids)
    do i = 1, n
       order(i) = i
    end do
    do i = n / 2, 1, -1
       call sift(i, n)
    end do
    do last = n, 2, -1
       tmp = order(1); order(1) = order(last); order(last) = tmp
       call sift(1, last - 1)
    end do
  contains
    subroutine sift(start, bottom)
      integer, intent(in) :: start, bottom
      root = start
      do while (2 * root <= bottom)
         child = 2 * root
         if (child < bottom) then
            if (ids(order(child)) < ids(order(child + 1))) child = child + 1
         end if
         if (ids(order(root)) >= ids(order(child))) return
         tmp = order(root); order(root) = order(child); order(child) = tmp
         root = child
      end do
    end subroutine sift
  end subroutine sort_index_by_id

  !> Download the resident cells and rebuild aero_states(:): the mirrored fields come from the
  !> device; the fields the device does not carry are re-attached to the particle with the
  !> same id (see the module header), reset when its composition changed, defaulted for a
  !> particle the host has not seen.  Weights come back too (a rebalance changes them).
  subroutine pmcx_download_states(aero_states)
    type(aero_state_t), intent(inout) :: aero_states(:)

    integer :: n_cells, n_spec, n_class, n_source, c, i, s, k, j, lo, hi, mid, n_old, n_have, n_add
    integer(c_int64_t) :: nt, o
    integer(c_int64_t), allocatable :: part_offset(:), id(:)
    real(c_double), allocatable :: vol(:,:), lct(:), gct(:), mag(:,:)
    integer(c_int32_t), allocatable, target :: n_orig(:,:)
    integer(c_int32_t), allocatable :: wclass(:), n_part(:)
    type(aero_particle_t), allocatable :: old(:), new(:)
    type(aero_component_t), allocatable :: comp(:)
    integer(kind=8), allocatable :: old_id(:)
    integer, allocatable :: order(:), have(:)
    logical, allocatable :: taken(:)
    logical :: changed
    real(kind=dp) :: t_new
    type(c_ptr) :: p_norig

    n_cells = cfg%n_cells
    n_spec = cfg%n_spec
    n_class = cfg%n_class
    n_source = cfg%n_source
    allocate(n_part(n_cells), part_offset(n_cells + 1), mag(n_class, n_cells))
    call pmcx_die(pmcx_get_n_part(ctx, n_part), 'pmcx_get_n_part')
    call pmcx_die(pmcx_get_weights(ctx, mag), 'pmcx_get_weights')
    part_offset(1) = 0
    do c = 1, n_cells
       part_offset(c + 1) = part_offset(c) + n_part(c)
    end do
    nt = part_offset(n_cells + 1)
    ! packed arrays are [n_spec][n_total]: the leading dimension is the total
    allocate(vol(max(nt, 1_c_int64_t), n_spec), wclass(max(nt, 1_c_int64_t)), &
         id(max(nt, 1_c_int64_t)), lct(max(nt, 1_c_int64_t)), gct(max(nt, 1_c_int64_t)))
    allocate(n_orig(max(nt, 1_c_int64_t), max(n_source, 1)))
    p_norig = c_null_ptr
    if (n_source > 0) p_norig = c_loc(n_orig)
    call pmcx_die(pmcx_download_cells(ctx, 0, n_cells, part_offset, vol, wclass, id, lct, gct, &
         p_norig), 'pmcx_download_cells')
    allocate(have(max(n_source, 1)))
    do c = 1, n_cells
       o = part_offset(c)
       ! the host's particles of this cell, findable by id
       n_old = 0
       if (allocated(aero_states(c)%apa%particle)) then
          n_old = min(aero_states(c)%apa%n_part, size(aero_states(c)%apa%particle))
          call move_alloc(aero_states(c)%apa%particle, old)
       end if
       if (allocated(old_id)) deallocate(old_id, order, taken)
       allocate(old_id(n_old), order(n_old), taken(n_old))
       do i = 1, n_old
          old_id(i) = old(i)%id
       end do
       call sort_index_by_id(old_id, order)
       taken = .false.
       allocate(new(n_part(c)))
       do i = 1, n_part(c)
          j = 0
          lo = 1; hi = n_old
          do while (lo <= hi)
             mid = (lo + hi) / 2
             if (old_id(order(mid)) < id(o + i)) then
                lo = mid + 1
             else if (old_id(order(mid)) > id(o + i)) then
                hi = mid - 1
             else
                j = order(mid); exit
             end if
          end do
          if (j > 0) then
             if (taken(j)) j = 0
          end if
          changed = .true.
          if (j > 0) then
             ! the same particle: what the device does not carry stays with it
             taken(j) = .true.
             changed = .false.
             do s = 1, n_spec
                if (old(j)%vol(s) /= vol(o + i, s)) changed = .true.
             end do
             if (allocated(old(j)%component)) call move_alloc(old(j)%component, new(i)%component)
             new(i)%weight_group = old(j)%weight_group
             new(i)%water_hyst_leg = old(j)%water_hyst_leg  ! kept by the merged particle [UPSTREAM-UNCITED]
             if (.not. changed) then
                if (allocated(old(j)%absorb_cross_sect)) call move_alloc(old(j)%absorb_cross_sect, new(i)%absorb_cross_sect)
                if (allocated(old(j)%scatter_cross_sect)) call move_alloc(old(j)%scatter_cross_sect, new(i)%scatter_cross_sect)
                if (allocated(old(j)%asymmetry)) call move_alloc(old(j)%asymmetry, new(i)%asymmetry)
                if (allocated(old(j)%refract_shell)) call move_alloc(old(j)%refract_shell, new(i)%refract_shell)
                if (allocated(old(j)%refract_core)) call move_alloc(old(j)%refract_core, new(i)%refract_core)
                new(i)%core_vol = old(j)%core_vol
             end if
          end if
          if (changed) then
             ! optics belong to a composition that no longer exists (or never did on this
             ! host): zero until MOSAIC computes them again (driver:1452-1466)
             allocate(new(i)%absorb_cross_sect(n_swbands), new(i)%scatter_cross_sect(n_swbands), &
                  new(i)%asymmetry(n_swbands), new(i)%refract_shell(n_swbands), &
                  new(i)%refract_core(n_swbands))
             new(i)%absorb_cross_sect = 0d0; new(i)%scatter_cross_sect = 0d0
             new(i)%asymmetry = 0d0
             new(i)%refract_shell = (0d0, 0d0); new(i)%refract_core = (0d0, 0d0)
             new(i)%core_vol = 0d0
          end if
          allocate(new(i)%vol(n_spec))
          do s = 1, n_spec
             new(i)%vol(s) = vol(o + i, s)
          end do
          new(i)%weight_class = wclass(o + i) + 1
          new(i)%id = id(o + i)
          ! component(:) <- the source histogram: constituents the particle had keep their
          ! create_time; the ones it gained are appended with the bound that moved
          if (n_source > 0) then
             have = 0
             n_have = aero_particle_n_components(new(i))
             do k = 1, n_have
                s = new(i)%component(k)%source_id
                if (s >= 1 .and. s <= n_source) have(s) = have(s) + 1
             end do
             n_add = 0
             do s = 1, n_source
                if (n_orig(o + i, s) < have(s)) n_have = -1   ! (cannot shrink: rebuild)
                n_add = n_add + max(0, n_orig(o + i, s) - have(s))
             end do
             if (n_have < 0) then
                if (allocated(new(i)%component)) deallocate(new(i)%component)
                have = 0; n_have = 0
                n_add = sum(n_orig(o + i, 1:n_source))
             end if
             if (n_add > 0) then
                allocate(comp(n_have + n_add))
                if (n_have > 0) comp(1:n_have) = new(i)%component(1:n_have)
                t_new = gct(o + i)
                if (j > 0) then
                   if (lct(o + i) < old(j)%least_create_time) t_new = lct(o + i)
                end if
                k = n_have
                do s = 1, n_source
                   do mid = 1, n_orig(o + i, s) - have(s)
                      k = k + 1
                      comp(k)%source_id = s
                      comp(k)%create_time = t_new
                   end do
                end do
                if (j == 0) comp(1)%create_time = lct(o + i)   ! a new particle: both bounds appear
                call move_alloc(comp, new(i)%component)
             end if
          end if
          new(i)%least_create_time = lct(o + i)
          new(i)%greatest_create_time = gct(o + i)
       end do
       if (allocated(old)) deallocate(old)
       call move_alloc(new, aero_states(c)%apa%particle)
       aero_states(c)%apa%n_part = n_part(c)
       do s = 1, n_class
          aero_states(c)%awa%weight(1, s)%magnitude = mag(s, c)
       end do
       aero_states(c)%valid_sort = .false.
    end do
  end subroutine pmcx_download_states

  !> mc_coag on the resident cells (interface/wrf_pmc_driver.F90:192-194 for every cell).
  subroutine mc_coag_resident_pmcx(coag_kernel_type, del_t, tot_n_samp, tot_n_coag)
    integer, intent(in) :: coag_kernel_type
    real(kind=dp), intent(in) :: del_t
    integer, intent(out) :: tot_n_samp(:), tot_n_coag(:)
    integer(c_int32_t), allocatable :: ns(:), nc(:), st(:)
    call pmcx_die(pmcx_mc_coag(ctx, coag_kernel_type, del_t, rng_seed, rng_step), 'pmcx_mc_coag')
    allocate(ns(cfg%n_cells), nc(cfg%n_cells), st(cfg%n_cells))
    call pmcx_die(pmcx_get_coag_counters(ctx, ns, nc, st), 'pmcx_get_coag_counters')
    if (any(iand(st, 3) /= 0)) call pmcx_die(5, 'mc_coag (event/append log overflow)')
    if (any(iand(st, 32) /= 0)) call pmcx_die(5, 'mc_coag (particle created above the kernel-bounds range)')
    tot_n_samp = ns
    tot_n_coag = nc
  end subroutine mc_coag_resident_pmcx

  !> The stochastic transport of wrf_pmc_trans_driver (interface/wrf_pmc_trans.F90:37-200) on
  !> the resident cells, ordered c = (j * nz + k) * nx + i (0-based).  prob_h(4, n_class, n_cells):
  !> east, west, north, south; prob_v(nz, n_class, n_cells): to each level of the column.
  subroutine trans_cells_pmcx(nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v, cell_volume, &
       n_moved, n_arrived)
    integer, intent(in) :: nx, nz, ny
    logical, intent(in) :: periodic_x, periodic_y
    real(kind=dp), intent(in), target :: prob_h(:,:,:), prob_v(:,:,:)
    real(kind=dp), intent(in) :: cell_volume(:)
    integer, intent(out) :: n_moved(:), n_arrived(:)
    integer(c_int32_t), allocatable :: nm(:), na(:)
    allocate(nm(cfg%n_cells), na(cfg%n_cells))
    call pmcx_die(pmcx_transport(ctx, nx, nz, ny, merge(1, 0, periodic_x), merge(1, 0, periodic_y), &
         c_loc(prob_h), c_loc(prob_v), cell_volume, rng_seed, rng_step, nm, na), 'pmcx_transport')
    n_moved = nm
    n_arrived = na
  end subroutine trans_cells_pmcx

  !> trans_aero_preweight (interface/wrf_pmc_trans_aero.F90:1103-1423) on the resident cells, with
  !> the probabilities handed over (as for trans_cells_pmcx).  expected_num_conc(n_class, n_cells).
  subroutine trans_preweight_cells_pmcx(nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v, &
       cell_volume, expected_num_conc, n_removed)
    integer, intent(in) :: nx, nz, ny
    logical, intent(in) :: periodic_x, periodic_y
    real(kind=dp), intent(in), target :: prob_h(:,:,:), prob_v(:,:,:)
    real(kind=dp), intent(in) :: cell_volume(:)
    real(kind=dp), intent(out) :: expected_num_conc(:,:)
    integer, intent(out) :: n_removed(:)
    integer(c_int32_t), allocatable :: nr(:)
    allocate(nr(cfg%n_cells))
    call pmcx_die(pmcx_transport_preweight(ctx, nx, nz, ny, merge(1, 0, periodic_x), &
         merge(1, 0, periodic_y), c_loc(prob_h), c_loc(prob_v), cell_volume, rng_seed, rng_step, &
         expected_num_conc, nr), 'pmcx_transport_preweight')
    n_removed = nr
  end subroutine trans_preweight_cells_pmcx

  !> The whole particle half of wrf_pmc_trans_driver (interface/wrf_pmc_trans.F90:95-200) from WRF's
  !> fields: probabilities built on the device, preweight, sampling -- nothing but the fields
  !> crosses the boundary.  The arrays are the tile's copies of grid%alt, %xkhh (one halo cell in
  !> x and y), %u_flux, %v_flux, %w_flux (class c in the last index), %z, %z_at_w, %exch_h in WRF's
  !> (i, k, j) order, single precision as the registry declares them.
  subroutine trans_from_fields_cells_pmcx(nx, nz, ny, periodic_x, periodic_y, dt, dx, dy, &
       vertmix_onoff, alt, xkhh, u_flux, v_flux, w_flux, z, z_at_w, exch_h, cell_volume, &
       n_moved, n_arrived)
    integer, intent(in) :: nx, nz, ny, vertmix_onoff
    logical, intent(in) :: periodic_x, periodic_y
    real(kind=dp), intent(in) :: dt, dx, dy
    real(c_float), intent(in), target :: alt(:,:,:), xkhh(:,:,:), u_flux(:,:,:,:), v_flux(:,:,:,:), &
         w_flux(:,:,:,:), z(:,:,:), z_at_w(:,:,:), exch_h(:,:,:)
    real(kind=dp), intent(in) :: cell_volume(:)
    integer, intent(out) :: n_moved(:), n_arrived(:)
    type(pmcx_trans_fields_t) :: f
    integer(c_int32_t), allocatable :: nm(:), na(:), nr(:)
    real(c_double), allocatable :: ex(:)
    f%nx = nx; f%nz = nz; f%ny = ny
    f%do_vertmix = vertmix_onoff
    f%all_columns = merge(1, 0, periodic_x .and. periodic_y)
    f%i_offset = 0; f%j_offset = 0; f%global_nx = nx; f%global_ny = ny
    f%dt = dt; f%dx = dx; f%dy = dy
    f%alt = c_loc(alt); f%xkhh = c_loc(xkhh); f%u_flux = c_loc(u_flux); f%v_flux = c_loc(v_flux)
    f%w_flux = c_loc(w_flux); f%z = c_loc(z); f%z_at_w = c_loc(z_at_w); f%exch_h = c_loc(exch_h)
    call pmcx_die(pmcx_transport_probs(ctx, f, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, &
         c_null_ptr), 'pmcx_transport_probs')
    allocate(nm(cfg%n_cells), na(cfg%n_cells), nr(cfg%n_cells), ex(cfg%n_cells * cfg%n_class))
    call pmcx_die(pmcx_transport_preweight(ctx, nx, nz, ny, merge(1, 0, periodic_x), &
         merge(1, 0, periodic_y), c_null_ptr, c_null_ptr, cell_volume, rng_seed, rng_step, ex, nr), &
         'pmcx_transport_preweight')
    call pmcx_die(pmcx_transport(ctx, nx, nz, ny, merge(1, 0, periodic_x), merge(1, 0, periodic_y), &
         c_null_ptr, c_null_ptr, cell_volume, rng_seed, rng_step, nm, na), 'pmcx_transport')
    n_moved = nm
    n_arrived = na
  end subroutine trans_from_fields_cells_pmcx

  !> seasalt_emissions (interface/wrf_pmc_driver.F90:212-223, 1767-1924) on the resident cells whose
  !> ocean_surface flag is set; i_na, i_cl: aero_data_spec_by_name (1-based).
  subroutine seasalt_cells_pmcx(dt, elapsed_time, sea_salt_opt, i_na, i_cl, ocean_surface, wind_speed, &
       height, mode_class, mode_source, n_emit)
    real(kind=dp), intent(in) :: dt, elapsed_time
    integer, intent(in) :: sea_salt_opt, i_na, i_cl
    integer, intent(in) :: ocean_surface(:), mode_class(:), mode_source(:)
    real(kind=dp), intent(in) :: wind_speed(:), height(:)
    integer, intent(out) :: n_emit(:)
    integer(c_int32_t), allocatable :: oc(:), mc(:), ms(:), ne(:)
    allocate(oc(cfg%n_cells), ne(cfg%n_cells), mc(size(mode_class)), ms(size(mode_source)))
    oc = ocean_surface
    mc = mode_class - 1          ! string_array_find is 1-based
    ms = mode_source - 1
    call pmcx_die(pmcx_seasalt_emissions(ctx, dt, elapsed_time, rng_seed, rng_step, sea_salt_opt, &
         i_na - 1, i_cl - 1, oc, wind_speed, height, mc, ms, ne, c_null_ptr), 'pmcx_seasalt_emissions')
    n_emit = ne
  end subroutine seasalt_cells_pmcx

  !> Dry deposition of wrf_pmc_dep_aero.F90:28-192 on the resident surface cells
  !> (surface(c) /= 0); the per-cell inputs are the ones the reference reads from grid%.
  subroutine dry_dep_cells_pmcx(dt, surface, aer_res_a, ustar, alpha, big_a, height, inv_density, &
       n_removed)
    real(kind=dp), intent(in) :: dt
    integer, intent(in) :: surface(:)
    real(kind=dp), intent(in) :: aer_res_a(:), ustar(:), alpha(:), big_a(:), height(:), inv_density(:)
    integer, intent(out) :: n_removed(:)
    integer(c_int32_t), allocatable :: sf(:), nr(:), bad(:)
    allocate(sf(cfg%n_cells), nr(cfg%n_cells), bad(cfg%n_cells))
    sf = surface
    call pmcx_die(pmcx_dry_deposition(ctx, dt, rng_seed, rng_step, sf, aer_res_a, ustar, alpha, &
         big_a, height, inv_density, nr, bad), 'pmcx_dry_deposition')
    n_removed = nr
  end subroutine dry_dep_cells_pmcx

  !> mc_coag on a flat array of cells: gather, upload, step, download, scatter.
  subroutine partmc_coag_cells_pmcx(coag_kernel_type, env_states, aero_data, aero_states, &
       del_t, tot_n_samp, tot_n_coag)
    integer, intent(in) :: coag_kernel_type
    type(env_state_t), intent(in) :: env_states(:)
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(inout) :: aero_states(:)
    real(kind=dp), intent(in) :: del_t
    integer, intent(out) :: tot_n_samp(:), tot_n_coag(:)
    call pmcx_upload_states(env_states, aero_data, aero_states)
    call mc_coag_resident_pmcx(coag_kernel_type, del_t, tot_n_samp, tot_n_coag)
    call pmcx_download_states(aero_states)
  end subroutine partmc_coag_cells_pmcx

  !> Drop-in for mc_coag (same argument list, one cell).
  subroutine mc_coag_pmcx(coag_kernel_type, env_state, aero_data, aero_state, del_t, &
       tot_n_samp, tot_n_coag)
    integer, intent(in) :: coag_kernel_type
    type(env_state_t), intent(in) :: env_state
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(inout) :: aero_state
    real(kind=dp), intent(in) :: del_t
    integer, intent(out) :: tot_n_samp, tot_n_coag
    type(env_state_t) :: envs(1)
    type(aero_state_t) :: states(1)
    integer :: ns(1), nc(1)
    envs(1) = env_state
    call move_alloc_state(aero_state, states(1))
    call partmc_coag_cells_pmcx(coag_kernel_type, envs, aero_data, states, del_t, ns, nc)
    call move_alloc_state(states(1), aero_state)
    tot_n_samp = ns(1)
    tot_n_coag = nc(1)
  end subroutine mc_coag_pmcx

  subroutine move_alloc_state(from, to)
    type(aero_state_t), intent(inout) :: from, to
    to%apa%n_part = from%apa%n_part
    call move_alloc(from%apa%particle, to%apa%particle)
    call move_alloc(from%awa%weight, to%awa%weight)
    if (allocated(from%n_part_ideal)) call move_alloc(from%n_part_ideal, to%n_part_ideal)
    to%valid_sort = from%valid_sort
  end subroutine move_alloc_state

  !> aero_state_rebalance on the RESIDENT cells (state stays on the device; INTEGRATION.md 2c).
  subroutine aero_state_rebalance_cells_pmcx(allow_doubling, allow_halving)
    logical, intent(in) :: allow_doubling, allow_halving
    integer(c_int32_t) :: d, h
    integer(c_int32_t), allocatable :: ns(:), nc(:), st(:)
    d = merge(1, 0, allow_doubling)
    h = merge(1, 0, allow_halving)
    call pmcx_die(pmcx_rebalance(ctx, d, h, rng_seed, rng_step), 'pmcx_rebalance')
    allocate(ns(cfg%n_cells), nc(cfg%n_cells), st(cfg%n_cells))
    call pmcx_die(pmcx_get_coag_counters(ctx, ns, nc, st), 'pmcx_get_coag_counters')
    if (any(iand(st, 16) /= 0)) call pmcx_die(5, 'aero_state_rebalance (a doubled cell exceeds its capacity)')
  end subroutine aero_state_rebalance_cells_pmcx

  !> aero_state_rebalance on host states (interface/wrf_pmc_driver.F90:248-250 for every
  !> cell): upload (with n_part_ideal), rebalance, download.  The form for a host that keeps
  !> aero_states(:) authoritative between calls (INTEGRATION.md 2b).
  subroutine aero_state_rebalance_pmcx(env_states, aero_data, aero_states, allow_doubling, allow_halving)
    type(env_state_t), intent(in) :: env_states(:)
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(inout) :: aero_states(:)
    logical, intent(in) :: allow_doubling, allow_halving
    call pmcx_upload_states(env_states, aero_data, aero_states)
    call aero_state_rebalance_cells_pmcx(allow_doubling, allow_halving)
    call pmcx_download_states(aero_states)
  end subroutine aero_state_rebalance_pmcx

end module pmcx_shim
