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
! id, least/greatest_create_time and component(:) -- as a per-source histogram and, with
! pmcx_shim_set_components(n) (n = 1..8, default 8), as the list itself: source_id and
! create_time of a particle's first n constituents travel with it on the device and
! component(:) comes back exactly for every particle of at most n constituents (DESIGN.md 3.2).
! Everything else a particle carries on the Fortran side (water_hyst_leg, weight_group, the
! optics arrays, core_vol) stays on the host and is re-attached BY ID after a download; a
! particle whose composition changed gets its optics reset, a particle the host has never seen
! (emission, transport arrival, doubling) starts from defaults.  Constituents beyond the n
! carried (or all of them with n = 0) are rebuilt from the histogram with a create_time bounded
! by the particle's least/greatest create time.
module pmcx_shim
  use iso_c_binding
  use pmcx_c           ! the C entry points of include/pmcx.h, every one bound (pmcx_c.F90)
  use pmc_mock         ! inside WRF: pmc_aero_state, pmc_aero_data, pmc_env_state, pmc_scenario, ...
  use wrf_mock         ! inside WRF: module_domain, module_configure, module_model_constants
  implicit none
  private
  ! the two drop-in drivers (the reference's exact argument lists): one partmc_timestep, one
  ! wrf_pmc_trans_driver per WRF step on the device-resident block; partmc_sync_host_pmcx
  ! brings aero_states / gas_states up to date when the Fortran side needs them (output, restart)
  public :: partmc_timestep_pmcx, wrf_pmc_trans_driver_pmcx, partmc_sync_host_pmcx, &
       partmc_process_pmcx, pmcx_shim_is_resident
  public :: pmcx_shim_set_components, pmcx_shim_set_headroom, pmcx_shim_set_auto_grow, pmcx_shim_growth
  public :: pmcx_shim_set_neighbours, pmcx_shift_t
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
       pmcx_device_alloc, pmcx_device_free, pmcx_device_copy, pmcx_transport_preweight_begin, &
       pmcx_transport_preweight_finish, pmcx_set_transport_lateral


  ! one context per (rank, cell count); rebuilt when the shape changes
  type(c_ptr), save :: ctx = c_null_ptr
  type(pmcx_config), save :: cfg
  logical, save :: have_ctx = .false.
  integer, save :: want_comp_max = 8           ! entries of component(:) carried per particle
  real(kind=dp), save :: headroom = 3.0d0      ! particle slots of a cell / its ideal count (pmcx_shim_set_headroom)
  logical, save :: auto_grow = .true.          ! the block grows when a cell runs short of slots (pmcx_shim_set_auto_grow)

  ! ---- a rank of WRF's patch decomposition (pmcx_shim_set_neighbours) ----
  !> How a rank reaches its neighbours: WRF supplies MPI_Sendrecv, a test a stand-in.  Send the n_send
  !> bytes at `send` to the neighbour on side dir (0 south, 1 north, 2 west, 3 east) and receive n_recv
  !> bytes into `recv` from the neighbour on the OPPOSITE side -- what it sent in the same direction
  !> (one shift of a halo exchange: MPI_Sendrecv(send, ..., dest = neighbour(dir), recv, ..., source =
  !> neighbour(opposite))).  n_send = 0 / n_recv = 0: no neighbour there, nothing travels.  Host memory.
  abstract interface
     subroutine pmcx_shift_t(dir, send, n_send, recv, n_recv)
       import :: c_ptr, c_int64_t
       integer, intent(in) :: dir
       type(c_ptr), value :: send, recv
       integer(c_int64_t), value :: n_send, n_recv
     end subroutine pmcx_shift_t
  end interface
  procedure(pmcx_shift_t), pointer, save :: shift_proc => null()
  logical, save :: has_nbr(0:3) = .false.      ! a neighbouring rank on side 0 south, 1 north, 2 west, 3 east
  logical, save :: in_padded = .false.         ! the drivers are running on the patch + its ghost ring
  integer, save :: real_is = 0, real_ie = 0, real_js = 0, real_je = 0   ! the patch inside the padded block
  integer, save :: pad_gnx = 0, pad_gny = 0    ! the domain's columns and rows (ids of a cell's particles)
  integer, save :: pad_nxb = 0, pad_nz = 0, pad_nyb = 0, pad_is0 = 0, pad_js0 = 0   ! the padded block: its shape, the
                                               ! (1-based, global) column and row of its first cell
  type(env_state_t), allocatable, save :: p_env(:,:,:)
  type(scenario_t), allocatable, save :: p_scn(:,:,:)
  type(aero_state_t), allocatable, save :: p_aero(:,:,:)
  type(gas_state_t), allocatable, save :: p_gas(:,:,:)
  ! per-step staging arrays in page-locked memory (pmcx_host_alloc): an upload from them is one
  ! DMA that overlaps the device's work; they grow as needed and live until pmcx_shim_finalize
  integer, parameter :: N_PIN = 24
  type(c_ptr), save :: pin_p(N_PIN) = c_null_ptr
  integer(c_int64_t), save :: pin_bytes(N_PIN) = 0
  integer(c_int64_t), save :: rng_seed = 0     ! grid%random_seed (wrf_pmc_init.F90:131)
  integer(c_int32_t), save :: rng_step = 0     ! chemistry-step index
  integer(c_int32_t), save :: cell_id_base = 0 ! global id of this rank's first cell
  integer(kind=8), parameter :: ID_BLOCK = 4294967296_8   ! ids a cell may create between uploads
  integer(kind=8), save :: id_block_base = 0              ! first id of cell 1's block
  ! the resident block of the two drivers
  logical, save :: resident = .false., have_gas = .false.
  integer, save :: res_nx = 0, res_nz = 0, res_ny = 0
  integer(kind=8), save, public :: last_n_samp = 0, last_n_coag = 0, last_n_moved = 0, last_n_deposited = 0
  ! depvel_partmc's aerodynamic resistance of the last deposition call, per cell (inspection)
  real(c_double), allocatable, save, public :: last_aer_res(:)
  ! scenario(i,k,j)'s aerosol modes at their current time levels, as sent to the device
  integer, save :: em_n_mode = -1, bg_n_mode = -1
  integer, allocatable, save :: em_level(:), bg_level(:)
  real(c_double), allocatable, save, target :: em_num(:,:), em_gm(:,:), em_gs(:,:), em_vf(:,:,:), em_scale(:)
  real(c_double), allocatable, save, target :: bg_num(:,:), bg_gm(:,:), bg_gs(:,:), bg_vf(:,:,:), bg_rate(:)
  integer(c_int32_t), allocatable, save, target :: em_cls(:,:), em_src(:,:), bg_cls(:,:), bg_src(:,:)

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

  !> How many entries of particle%component(:) the device carries per particle (0..8; 0: only
  !> the source histogram).  Takes effect at the next upload that builds a context.
  subroutine pmcx_shim_set_components(n)
    integer, intent(in) :: n
    want_comp_max = max(0, min(8, n))
  end subroutine pmcx_shim_set_components

  !> Particle slots per cell as a multiple of the ideal count (default 3).  The
  !> reference's particle arrays grow; a cell here has its slots for the whole run: the rebalance
  !> keeps a class below twice its ideal count at the END of a step, inside a step (arrivals
  !> before the rebalance, 1/ratio copies per arrival, a boundary refill) it can be above.  A run
  !> that stops with "a cell ran out of particle slots" wants a larger factor; it costs memory only
  !> (about 100 bytes a slot with one species; a domain that fills the card wants a smaller one).
  !> Takes effect at the next upload that builds a context.
  subroutine pmcx_shim_set_headroom(factor)
    real(kind=dp), intent(in) :: factor
    headroom = max(1.0d0, factor)
  end subroutine pmcx_shim_set_headroom

  !> Whether the block grows by itself when a cell runs short of slots (default .true.: PartMC's
  !> particle arrays grow on use).  .false.: a full cell stops the run with its message, as until
  !> round 4.  Takes effect at the next upload that builds a context.
  subroutine pmcx_shim_set_auto_grow(on)
    logical, intent(in) :: on
    auto_grow = on
  end subroutine pmcx_shim_set_auto_grow

  !> This rank holds a patch of a domain cut across ranks (WRF's decomposition in x and y,
  !> WRFV3/test/em_real/namelist.input.cares_bw_full:57-58; the exchanges of interface/
  !> wrf_pmc_trans_aero.F90:148-273, 381-473, 543-675, 1261-1343): `shift` moves bytes between
  !> neighbouring ranks, has_* say on which sides a neighbour exists (across a periodic edge of the
  !> domain counts; the domain's open edge does not).  From then on partmc_timestep_pmcx and
  !> wrf_pmc_trans_driver_pmcx -- called with the patch's bounds, as the reference's drivers are --
  !> keep a resident block of the patch plus one ghost row and column on every side, and the transport
  !> driver exchanges the preweight's terms, the boundary strips' weights and the movers with the four
  !> neighbours: the result on the patch is the whole domain's, bit for bit.  Call before the first step.
  subroutine pmcx_shim_set_neighbours(shift, has_south, has_north, has_west, has_east)
    procedure(pmcx_shift_t) :: shift
    logical, intent(in) :: has_south, has_north, has_west, has_east
    shift_proc => shift
    has_nbr = [has_south, has_north, has_west, has_east]
  end subroutine pmcx_shim_set_neighbours

  !> v modulo 2**32 as the 32-bit pattern C reads as uint32_t
  integer(c_int32_t) function u32(v)
    integer(kind=8), intent(in) :: v
    integer(kind=8) :: w
    w = modulo(v, 4294967296_8)
    if (w >= 2147483648_8) w = w - 4294967296_8
    u32 = int(w, c_int32_t)
  end function u32

  !> The padded arrays of a rank of a cut domain: the patch plus one ghost row and column on every side.
  !> A ghost cell holds no particles; its weights, ideal counts, scenario and gases are copies of the
  !> nearest cell of the patch (weights: until the exchange brings its owner's), its meteorology comes
  !> from grid%'s halo like any cell's.  Built on the first call; later calls keep p_env (the old
  !> heights) and only note the patch's bounds.
  subroutine pad_states(grid, scenario, env_states, aero_states, gas_states, is, ie, js, je, ks, ke, gnx, gny)
    type(domain), intent(in) :: grid
    integer, intent(in) :: is, ie, js, je, ks, ke, gnx, gny
    type(scenario_t), intent(in) :: scenario(is:ie, ks:ke, js:je)
    type(env_state_t), intent(in) :: env_states(is:ie, ks:ke, js:je)
    type(aero_state_t), intent(in) :: aero_states(is:ie, ks:ke, js:je)
    type(gas_state_t), intent(in) :: gas_states(is:ie, ks:ke, js:je)
    integer :: i, j, k, i0, j0
    real_is = is; real_ie = ie; real_js = js; real_je = je
    pad_gnx = gnx; pad_gny = gny
    pad_nxb = ie - is + 3; pad_nz = ke - ks + 1; pad_nyb = je - js + 3; pad_is0 = is - 1; pad_js0 = js - 1
    if (allocated(p_env)) return
    allocate(p_env(is-1:ie+1, ks:ke, js-1:je+1), p_scn(is-1:ie+1, ks:ke, js-1:je+1), &
         p_aero(is-1:ie+1, ks:ke, js-1:je+1), p_gas(is-1:ie+1, ks:ke, js-1:je+1))
    do j = js - 1, je + 1
    do k = ks, ke
    do i = is - 1, ie + 1
       i0 = min(max(i, is), ie); j0 = min(max(j, js), je)
       p_env(i,k,j) = env_states(i0,k,j0)
       p_scn(i,k,j) = scenario(i0,k,j0)
       p_gas(i,k,j) = gas_states(i0,k,j0)
       if (i == i0 .and. j == j0) then
          p_aero(i,k,j) = aero_states(i,k,j)
       else
          p_aero(i,k,j)%awa = aero_states(i0,k,j0)%awa
          if (allocated(aero_states(i0,k,j0)%n_part_ideal)) p_aero(i,k,j)%n_part_ideal = aero_states(i0,k,j0)%n_part_ideal
          p_aero(i,k,j)%apa%n_part = 0
          allocate(p_aero(i,k,j)%apa%particle(0))
       end if
    end do
    end do
    end do
  end subroutine pad_states

  !> The block of 2**32 ids a cell's new particles come from: its number within the rank's block or --
  !> a rank of a cut domain -- within the whole domain, so that ids do not depend on the decomposition.
  integer function id_block_of(c)
    integer, intent(in) :: c
    integer :: i, k, j, gi, gj
    if (.not. in_padded) then
       id_block_of = c - 1
       return
    end if
    i = mod(c - 1, pad_nxb); k = mod((c - 1) / pad_nxb, pad_nz); j = (c - 1) / (pad_nxb * pad_nz)
    gi = modulo(pad_is0 + i - 1, pad_gnx); gj = modulo(pad_js0 + j - 1, pad_gny)
    id_block_of = (gj * pad_nz + k) * pad_gnx + gi
  end function id_block_of

  !> what every side sends goes to its neighbour; recv(s) takes what the neighbour on side s sent
  !> towards this rank (n_recv(s) bytes, known to both)
  subroutine swap4(send, n_send, recv, n_recv)
    type(c_ptr), intent(in) :: send(0:3), recv(0:3)
    integer(c_int64_t), intent(in) :: n_send(0:3), n_recv(0:3)
    integer :: dir, opp
    do dir = 0, 3
       opp = ieor(dir, 1)
       call shift_proc(dir, send(dir), merge(n_send(dir), 0_c_int64_t, has_nbr(dir)), &
            recv(opp), merge(n_recv(opp), 0_c_int64_t, has_nbr(opp)))
    end do
  end subroutine swap4

  !> Slots per cell now, and how many times the block has grown since it was made.
  subroutine pmcx_shim_growth(cap, n_grown)
    integer, intent(out) :: cap, n_grown
    integer(c_int32_t) :: c_, n_
    cap = 0; n_grown = 0
    if (.not. have_ctx) return
    call pmcx_die(pmcx_get_growth(ctx, c_, n_), 'pmcx_get_growth')
    cap = c_; n_grown = n_
  end subroutine pmcx_shim_growth

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
    cfg%n_comp_max = 0
    if (n_source >= 1 .and. n_source <= 255) cfg%n_comp_max = want_comp_max
    cfg%i_water = aero_data%i_water - 1
    cfg%cell_id_base = cell_id_base
    call pmcx_die(pmcx_create(cfg, ctx), 'pmcx_create')
    have_ctx = .true.
    ! PartMC's particle arrays grow on use (behind aero_state_add_particle, wrf_pmc_trans_aero.F90:777-804):
    ! the block grows by itself -- before a transport gathers arrivals a cell would not hold, and
    ! when the fullest cell passes 3/4 of its slots at the start of a step (pmcx_set_auto_grow)
    if (auto_grow) call pmcx_die(pmcx_set_auto_grow(ctx, 1, 0.75d0, 1.5d0), 'pmcx_set_auto_grow')
    call pmcx_die(pmcx_set_aero_data(ctx, aero_data%density, c_null_ptr, c_null_ptr), &
         'pmcx_set_aero_data')
  end subroutine pmcx_shim_init

  !> Page-locked staging buffer number `slot`, at least `bytes` long.
  function pinned(slot, bytes) result(p)
    integer, intent(in) :: slot
    integer(c_int64_t), intent(in) :: bytes
    type(c_ptr) :: p
    if (pin_bytes(slot) < bytes .or. .not. c_associated(pin_p(slot))) then
       if (c_associated(pin_p(slot))) call pmcx_die(pmcx_host_free(ctx, pin_p(slot)), 'pmcx_host_free')
       call pmcx_die(pmcx_host_alloc(ctx, max(bytes, 8_c_int64_t), pin_p(slot)), 'pmcx_host_alloc')
       pin_bytes(slot) = bytes
    end if
    p = pin_p(slot)
  end function pinned

  subroutine pmcx_shim_finalize()
    integer(c_int) :: rc
    integer :: q
    if (have_ctx) then
       do q = 1, N_PIN
          if (c_associated(pin_p(q))) rc = pmcx_host_free(ctx, pin_p(q))
       end do
    end if
    pin_p = c_null_ptr
    pin_bytes = 0
    if (have_ctx) rc = pmcx_destroy(ctx)
    have_ctx = .false.
    ctx = c_null_ptr
    resident = .false.
    em_n_mode = -1; bg_n_mode = -1
    if (allocated(em_level)) deallocate(em_level, em_num, em_gm, em_gs, em_vf, em_cls, em_src, em_scale)
    if (allocated(bg_level)) deallocate(bg_level, bg_num, bg_gm, bg_gs, bg_vf, bg_cls, bg_src, bg_rate)
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
    call upload_states_n(size(aero_states), env_states, aero_data, aero_states, cap_min)
  end subroutine pmcx_upload_states

  !> The same with the number of cells explicit: aero_states / env_states of any rank in memory
  !> order (the (i,k,j) arrays of the reference are the cells c = (j nz + k) nx + i in sequence).
  subroutine upload_states_n(n_cells, env_states, aero_data, aero_states, cap_min)
    integer, intent(in) :: n_cells
    type(env_state_t), intent(in) :: env_states(n_cells)
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(in) :: aero_states(n_cells)
    integer, intent(in), optional :: cap_min

    integer :: n_spec, n_class, n_source, c, i, s, k, cap_needed, cap_max
    integer(c_int64_t) :: nt, o
    integer(c_int64_t), allocatable :: part_offset(:), id(:)
    real(c_double), allocatable :: vol(:,:), lct(:), gct(:), temp(:), pres(:), rh(:), mag(:,:), &
         n_ideal(:,:)
    integer(c_int32_t), allocatable, target :: n_orig(:,:)
    integer(c_int32_t), allocatable :: wclass(:)
    type(c_ptr) :: p_norig
    logical :: have_ideal

    n_spec = size(aero_data%density)
    n_class = size(aero_states(1)%awa%weight, 2)
    n_source = aero_data_n_source(aero_data)
    cap_needed = 64
    if (present(cap_min)) cap_needed = max(cap_needed, cap_min)
    have_ideal = .true.
    do c = 1, n_cells
       cap_needed = max(cap_needed, aero_states(c)%apa%n_part)
       if (allocated(aero_states(c)%n_part_ideal)) then
          cap_needed = max(cap_needed, ceiling(headroom * sum(aero_states(c)%n_part_ideal(1, :))))
       else
          have_ideal = .false.
       end if
    end do
    ! list entries are 16-bit on the device: at most 65 280 slots per cell, whatever the number of
    ! weight classes (pmcx_max_cap)
    block
      type(pmcx_config) :: probe
      call pmcx_config_default(probe)
      probe%n_class = n_class
      cap_max = pmcx_max_cap(probe)
    end block
    cap_needed = ((cap_needed + 63) / 64) * 64
    if (cap_needed > cap_max) then
       write(0,'(a,i0,a,i0,a)') 'pmcx_shim: a cell needs ', cap_needed, ' particle slots (its particles, or the &
            &headroom times its n_part_ideal); the device holds at most ', cap_max, ' per cell (16-bit list entries)'
       call pmcx_die(1, 'pmcx_upload_states')
    end if
    if (have_ctx) then   ! (the block may have grown since it was made)
       block
         integer(c_int32_t) :: cap_now, n_grown
         call pmcx_die(pmcx_get_growth(ctx, cap_now, n_grown), 'pmcx_get_growth')
         cfg%cap = cap_now
       end block
    end if
    if (.not. have_ctx .or. cfg%n_cells /= n_cells .or. cfg%cap < cap_needed &
         .or. cfg%n_spec /= n_spec .or. cfg%n_class /= n_class .or. cfg%n_source /= n_source &
         .or. cfg%n_comp_max /= merge(want_comp_max, 0, n_source >= 1 .and. n_source <= 255)) then
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
    ! ids of particles created on the device (doubling, emission, transport copies, boundary
    ! copies): every cell draws from its own block of 2**32 ids of PartMC's counter, which moves
    ! past all of them; pmcx_download_states checks that no cell left its block
    block
      integer(c_int64_t), allocatable :: nid(:)
      allocate(nid(n_cells))
      id_block_base = next_id
      do c = 1, n_cells
         nid(c) = next_id + int(id_block_of(c), 8) * ID_BLOCK
      end do
      call pmcx_die(pmcx_set_next_id(ctx, nid), 'pmcx_set_next_id')
      next_id = next_id + int(merge(pad_gnx * pad_gny * pad_nz, n_cells, in_padded), 8) * ID_BLOCK
    end block
    call pmcx_die(pmcx_upload_cells(ctx, 0, n_cells, part_offset, vol, wclass, id, lct, gct, &
         p_norig), 'pmcx_upload_cells')
    ! component(:) itself, the first n_comp_max entries of every particle (driver:1421-1437)
    if (cfg%n_comp_max > 0) then
       block
         integer(c_int32_t), allocatable :: clen(:), csrc(:,:)
         real(c_double), allocatable :: ctim(:,:)
         integer :: kc, nk
         kc = cfg%n_comp_max
         allocate(clen(max(nt, 1_c_int64_t)), csrc(kc, max(nt, 1_c_int64_t)), ctim(kc, max(nt, 1_c_int64_t)))
         clen = 0; csrc = -1; ctim = 0d0
         do c = 1, n_cells
            o = part_offset(c)
            do i = 1, aero_states(c)%apa%n_part
               nk = 0
               do k = 1, aero_particle_n_components(aero_states(c)%apa%particle(i))
                  s = aero_states(c)%apa%particle(i)%component(k)%source_id
                  if (s < 1 .or. s > n_source) cycle   ! (as the histogram: not a source of aero_data)
                  if (nk >= kc) exit
                  nk = nk + 1
                  csrc(nk, o + i) = s - 1
                  ctim(nk, o + i) = aero_states(c)%apa%particle(i)%component(k)%create_time
               end do
               clen(o + i) = nk
            end do
         end do
         call pmcx_die(pmcx_upload_components(ctx, 0, n_cells, part_offset, clen, csrc, ctim), &
              'pmcx_upload_components')
       end block
    end if
  end subroutine upload_states_n

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
    if (size(aero_states) /= cfg%n_cells) call pmcx_die(1, 'pmcx_download_states (array size /= resident cells)')
    call download_states_n(size(aero_states), aero_states)
  end subroutine pmcx_download_states

  subroutine download_states_n(n_cells, aero_states)
    integer, intent(in) :: n_cells
    type(aero_state_t), intent(inout) :: aero_states(n_cells)

    integer :: n_spec, n_class, n_source, c, i, s, k, j, lo, hi, mid, n_old, n_have, n_add
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
    integer(c_int32_t), allocatable :: clen(:), csrc(:,:)
    real(c_double), allocatable :: ctim(:,:)
    integer :: kc

    n_spec = cfg%n_spec
    n_class = cfg%n_class
    n_source = cfg%n_source
    allocate(n_part(n_cells), part_offset(n_cells + 1), mag(n_class, n_cells))
    ! ids created on the device: every cell inside its block, or ids are no longer unique
    block
      integer(c_int64_t), allocatable :: nid(:)
      allocate(nid(n_cells))
      call pmcx_die(pmcx_get_next_id(ctx, nid), 'pmcx_get_next_id')
      do c = 1, n_cells
         if (nid(c) > id_block_base + int(id_block_of(c) + 1, 8) * ID_BLOCK) &
              call pmcx_die(5, 'pmcx_download_states (a cell created more than 2**32 particle ids)')
      end do
    end block
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
    kc = cfg%n_comp_max
    if (kc > 0) then
       allocate(clen(max(nt, 1_c_int64_t)), csrc(kc, max(nt, 1_c_int64_t)), ctim(kc, max(nt, 1_c_int64_t)))
       call pmcx_die(pmcx_download_components(ctx, 0, n_cells, part_offset, clen, csrc, ctim), &
            'pmcx_download_components')
    end if
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
          if (n_source > 0 .and. kc > 0) then
             ! the list travelled with the particle: its first clen entries are exact; those a
             ! particle of more than n_comp_max constituents has beyond them follow from the
             ! histogram, created within the particle's create-time bounds
             if (allocated(new(i)%component)) deallocate(new(i)%component)
             n_add = sum(n_orig(o + i, 1:n_source))
             allocate(comp(max(n_add, int(clen(o + i)))))
             have = 0
             do k = 1, clen(o + i)
                comp(k)%source_id = csrc(k, o + i) + 1
                comp(k)%create_time = ctim(k, o + i)
                have(csrc(k, o + i) + 1) = have(csrc(k, o + i) + 1) + 1
             end do
             k = clen(o + i)
             do s = 1, n_source
                do mid = 1, n_orig(o + i, s) - have(s)
                   k = k + 1
                   comp(k)%source_id = s
                   comp(k)%create_time = gct(o + i)
                end do
             end do
             call move_alloc(comp, new(i)%component)
          else if (n_source > 0) then
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
  end subroutine download_states_n

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

  ! ===========================================================================================
  ! The two drop-in drivers.  One line changes in WRFV3/frame/module_integrate.F:409-421:
  !     call partmc_timestep_pmcx(grid, config_flags, scenario, env_states, aero_data, aero_states, &
  !          gas_data, gas_states, pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke, global_nx, &
  !          global_ny, global_nz)
  !     call wrf_pmc_trans_driver_pmcx(grid, config_flags, env_states, scenario, aero_data, &
  !          aero_states, gas_states, pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke, global_nx, &
  !          global_ny, global_nz)
  ! The rank's block of cells lives on the device from the first call on; aero_states / gas_states
  ! on the host are stale until partmc_sync_host_pmcx (output and restart times).  One rank owning
  ! the whole domain is what these two cover; a domain cut across ranks uses the slab entry points
  ! (pmcx_transport_begin .. _finish, pmcx_transport_preweight_begin / _finish) underneath.
  ! ===========================================================================================

  !> Whether the block is on the device (partmc_timestep_pmcx uploads it on its first call).
  logical function pmcx_shim_is_resident()
    pmcx_shim_is_resident = resident
  end function pmcx_shim_is_resident

  !> wrf_to_partmc (interface/wrf_pmc_driver.F90:576-644): env_states from WRF's fields, the same
  !> expressions; the previous heights are what scenario_update_gas_state wants of old_env_states.
  subroutine wrf_to_partmc_pmcx(grid, nx, nz, ny, is, ks, js, env_states, old_height)
    type(domain), intent(in) :: grid
    integer, intent(in) :: nx, nz, ny, is, ks, js
    type(env_state_t), intent(inout) :: env_states(nx, nz, ny)
    real(kind=dp), intent(out) :: old_height(nx, nz, ny)
    integer :: i, j, k, gi, gj, gk
    real(kind=dp) :: pressure, potential_temp, temperature, rh, dx, dy, dz
    do j = 1, ny
    do k = 1, nz
    do i = 1, nx
       gi = is + i - 1; gk = ks + k - 1; gj = js + j - 1
       old_height(i,k,j) = env_states(i,k,j)%height                       ! driver:615
       pressure = grid%p(gi,gk,gj) + grid%pb(gi,gk,gj)                    ! driver:617
       potential_temp = grid%t_2(gi,gk,gj) + t0                           ! driver:619
       temperature = potential_temp * (pressure / p0)**(rcp)              ! driver:620
       env_states(i,k,j)%pressure = pressure
       env_states(i,k,j)%temp = temperature
       ! rel_hum, driver:649-666 (chem_prep's formula, clamped to 0.1 .. 0.95)
       rh = real(grid%moist(gi,gk,gj,p_qv), dp) / (3.8d0 * exp(17.27 * (temperature - 273.0d0) &
            / (temperature - 36.0d0)) / (.01 * pressure))
       env_states(i,k,j)%rel_humid = max(.1d0, min(rh, .95d0))
       env_states(i,k,j)%height = grid%z_at_w(gi,gk+1,gj) - grid%z_at_w(gi,gk,gj)   ! driver:627
       env_states(i,k,j)%altitude = grid%z(gi,gk,gj)
       env_states(i,k,j)%z_min = grid%z_at_w(gi,gk,gj)
       env_states(i,k,j)%z_max = grid%z_at_w(gi,gk+1,gj)
       env_states(i,k,j)%inverse_density = grid%alt(gi,gk,gj)             ! driver:633
       ! get_grid_cell_volume, driver:2002-2020
       dz = real(grid%z_at_w(gi,gk+1,gj) - grid%z_at_w(gi,gk,gj), kind=dp)
       dx = real((1.0d0 / grid%rdx) / grid%msftx(gi,gj), kind=dp)
       dy = real((1.0d0 / grid%rdy) / grid%msfty(gi,gj), kind=dp)
       env_states(i,k,j)%cell_volume = dx * dy * dz
       env_states(i,k,j)%cell_ix = gi; env_states(i,k,j)%cell_iy = gj; env_states(i,k,j)%cell_iz = gk
    end do
    end do
    end do
  end subroutine wrf_to_partmc_pmcx

  !> The cells of the reference's loop nests over the interior (wrf_pmc_driver.F90:157-171,
  !> wrf_pmc_trans_aero.F90:497-516): 1 inside, 0 for the lateral boundary cells of a domain that
  !> is not doubly periodic.
  subroutine interior_cells(config_flags, nx, nz, ny, is, js, global_nx, global_ny, inside)
    type(grid_config_rec_type), intent(in) :: config_flags
    integer, intent(in) :: nx, nz, ny, is, js, global_nx, global_ny
    integer(c_int32_t), intent(out) :: inside(nx, nz, ny)
    integer :: i, j, gi, gj
    inside = 1
    if (in_padded) then       ! the ghost ring of a rank's block: not this rank's cells
       inside(1,:,:) = 0; inside(nx,:,:) = 0; inside(:,:,1) = 0; inside(:,:,ny) = 0
    end if
    if (config_flags%periodic_x .and. config_flags%periodic_y) return
    do j = 1, ny
       do i = 1, nx
          gi = is + i - 1; gj = js + j - 1
          if (gi < 2 .or. gi > global_nx - 1 .or. gj < 2 .or. gj > global_ny - 1) inside(i,:,j) = 0
       end do
    end do
  end subroutine interior_cells

  !> aero_dist_interp_1d [upstream, uncited: a dist is held constant from a time point to the
  !> next; before the first point the first, from the last on the last]: the index into the list.
  integer function dist_level(n, times, t)
    integer, intent(in) :: n
    real(kind=dp), intent(in) :: times(n), t
    dist_level = max(1, min(find_1d(n, times, t), n))
  end function dist_level

  !> Gathers the modes of one kind (which = 0: aero_emission, 1: aero_background) of every cell's
  !> scenario at its current time level into the flat per-cell arrays of a pmcx_aero_dist
  !> (per_cell = 1); a cell without such a series gets modes of zero concentration.  level(c) is
  !> the time level each cell's arrays were filled for: the (large) mode arrays are rebuilt, and
  !> sent to the device again, only when some cell's level has changed.
  subroutine scenario_modes(which, n_cells, scenario, t, n_spec, n_mode, level, changed, &
       num_conc, gm, gs, vf, cls, src, rate)
    integer, intent(in) :: which, n_cells, n_spec
    type(scenario_t), intent(in) :: scenario(n_cells)
    real(kind=dp), intent(in) :: t
    integer, intent(inout) :: n_mode
    integer, allocatable, intent(inout) :: level(:)
    logical, intent(out) :: changed
    real(c_double), allocatable, intent(inout) :: num_conc(:,:), gm(:,:), gs(:,:), vf(:,:,:), rate(:)
    integer(c_int32_t), allocatable, intent(inout) :: cls(:,:), src(:,:)
    integer :: c, m, p, nt, nm, s
    nm = 0
    do c = 1, n_cells
       if (which == 0) then
          if (allocated(scenario(c)%aero_emission)) then
             if (size(scenario(c)%aero_emission) > 0) nm = max(nm, size(scenario(c)%aero_emission(1)%mode))
          end if
       else
          if (allocated(scenario(c)%aero_background)) then
             if (size(scenario(c)%aero_background) > 0) nm = max(nm, size(scenario(c)%aero_background(1)%mode))
          end if
       end if
    end do
    changed = .false.
    if (nm /= n_mode .or. .not. allocated(level)) then
       n_mode = nm
       if (allocated(level)) deallocate(level, num_conc, gm, gs, vf, cls, src, rate)
       allocate(level(n_cells), num_conc(max(nm, 1), n_cells), gm(max(nm, 1), n_cells), gs(max(nm, 1), n_cells), &
            vf(n_spec, max(nm, 1), n_cells), cls(max(nm, 1), n_cells), src(max(nm, 1), n_cells), rate(n_cells))
       level = -1
       changed = .true.
    end if
    if (nm == 0) return
    do c = 1, n_cells
       nt = 0
       if (which == 0) then
          if (allocated(scenario(c)%aero_emission_time)) nt = size(scenario(c)%aero_emission_time)
       else
          if (allocated(scenario(c)%aero_dilution_time)) nt = size(scenario(c)%aero_dilution_time)
       end if
       p = 0
       if (nt > 0) then
          if (which == 0) then
             p = dist_level(nt, scenario(c)%aero_emission_time, t)
          else
             p = dist_level(nt, scenario(c)%aero_dilution_time, t)
          end if
       end if
       if (p /= level(c)) then
          changed = .true.
          level(c) = p
          ! (-1: the cell's distribution does not have the mode -- nothing is drawn there and no
          ! weight is prepared for it, as when the reference loops over the modes the cell has)
          num_conc(:, c) = -1d0; gm(:, c) = 1d0; gs(:, c) = 0.1d0; vf(:, :, c) = 0d0; vf(1, :, c) = 1d0
          cls(:, c) = 0; src(:, c) = 0
          if (p > 0) then
             if (which == 0) then
                call fill(scenario(c)%aero_emission(p))
             else
                call fill(scenario(c)%aero_background(p))
             end if
          end if
       end if
       rate(c) = 0d0
       if (p > 0) then
          if (which == 0) then
             rate(c) = scenario(c)%aero_emission_rate_scale(p)
          else
             rate(c) = scenario(c)%aero_dilution_rate(p)
          end if
       end if
    end do
  contains
    subroutine fill(dist)
      type(aero_dist_t), intent(in) :: dist
      do m = 1, size(dist%mode)
         num_conc(m, c) = dist%mode(m)%num_conc
         gm(m, c) = dist%mode(m)%char_radius
         gs(m, c) = dist%mode(m)%log10_std_dev_radius
         do s = 1, n_spec
            vf(s, m, c) = dist%mode(m)%vol_frac(s)
         end do
         cls(m, c) = dist%mode(m)%weight_class - 1
         src(m, c) = dist%mode(m)%source - 1
      end do
    end subroutine fill
  end subroutine scenario_modes

  !> Uploads the rank's block (particles, weights, ideal counts, id blocks, gas states and the gas
  !> half of the scenarios) and names the cells of the interior loops.
  subroutine upload_domain(config_flags, scenario, env_states, aero_data, aero_states, gas_data, gas_states, &
       nx, nz, ny, is, js, global_nx, global_ny, cap_min)
    type(grid_config_rec_type), intent(in) :: config_flags
    integer, intent(in) :: nx, nz, ny, is, js, global_nx, global_ny, cap_min
    type(scenario_t), intent(in) :: scenario(nx * nz * ny)
    type(env_state_t), intent(in) :: env_states(nx * nz * ny)
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), intent(in) :: aero_states(nx * nz * ny)
    type(gas_data_t), intent(in) :: gas_data
    type(gas_state_t), intent(in) :: gas_states(nx * nz * ny)
    integer :: n_cells, n_gas, c, q, nte, ntd, k
    real(c_double), allocatable, target :: mix(:,:), et(:), es(:), em(:,:,:), dtm(:), dr(:,:), bg(:,:,:)
    integer(c_int32_t), allocatable, target :: inside(:)
    n_cells = nx * nz * ny
    call upload_states_n(n_cells, env_states, aero_data, aero_states, cap_min)
    allocate(inside(n_cells))
    call interior_cells(config_flags, nx, nz, ny, is, js, global_nx, global_ny, inside)
    if (all(inside == 1)) then
       call pmcx_die(pmcx_set_loop_cells(ctx, c_null_ptr), 'pmcx_set_loop_cells')
    else
       call pmcx_die(pmcx_set_loop_cells(ctx, c_loc(inside)), 'pmcx_set_loop_cells')
    end if
    ! gas_states(i,k,j)%mix_rat and the gas half of scenario(i,k,j) (wrf_pmc_init.F90:646-657,
    ! 718-720, 789-791, 901-919): resident from here on.  The time axes are those of the first cell
    ! that has them (the reference reads the same axes into every cell's scenario).
    n_gas = gas_data_n_spec(gas_data)
    have_gas = n_gas > 0
    if (.not. have_gas) return
    allocate(mix(n_gas, n_cells))
    do c = 1, n_cells
       mix(:, c) = gas_states(c)%mix_rat(1:n_gas)
    end do
    call pmcx_die(pmcx_set_gas_state(ctx, n_gas, mix), 'pmcx_set_gas_state')
    nte = 0; ntd = 0
    do c = 1, n_cells
       if (allocated(scenario(c)%gas_emission_time)) then
          if (nte == 0 .and. size(scenario(c)%gas_emission_time) > 0) then
             nte = size(scenario(c)%gas_emission_time)
             et = scenario(c)%gas_emission_time; es = scenario(c)%gas_emission_rate_scale
          end if
       end if
       if (allocated(scenario(c)%gas_dilution_time)) then
          if (ntd == 0 .and. size(scenario(c)%gas_dilution_time) > 0) then
             ntd = size(scenario(c)%gas_dilution_time)
             dtm = scenario(c)%gas_dilution_time
          end if
       end if
    end do
    if (nte > 0) then
       allocate(em(n_gas, nte, n_cells))
       em = 0d0
       do c = 1, n_cells
          if (.not. allocated(scenario(c)%gas_emission_time)) cycle
          if (size(scenario(c)%gas_emission_time) /= nte) cycle     ! (init_zero_emissions: no source here)
          do k = 1, nte
             em(:, k, c) = scenario(c)%gas_emission(k)%mix_rat(1:n_gas)
          end do
       end do
    end if
    if (ntd > 0) then
       allocate(dr(ntd, n_cells), bg(n_gas, ntd, n_cells))
       dr = 0d0; bg = 0d0
       do c = 1, n_cells
          if (.not. allocated(scenario(c)%gas_dilution_time)) cycle
          if (size(scenario(c)%gas_dilution_time) /= ntd) cycle
          dr(:, c) = scenario(c)%gas_dilution_rate
          do k = 1, ntd
             if (allocated(scenario(c)%gas_background(k)%mix_rat)) &
                  bg(:, k, c) = scenario(c)%gas_background(k)%mix_rat(1:n_gas)
          end do
       end do
    end if
    q = 0
    call pmcx_die(pmcx_set_gas_scenario(ctx, nte, ptr1(et, nte), ptr1(es, nte), ptr3(em, nte), &
         ntd, ptr1(dtm, ntd), ptr2(dr, ntd), ptr3(bg, ntd)), 'pmcx_set_gas_scenario')
  contains
    type(c_ptr) function ptr1(a, n)
      real(c_double), allocatable, target, intent(in) :: a(:)
      integer, intent(in) :: n
      ptr1 = c_null_ptr
      if (n > 0) ptr1 = c_loc(a)
    end function ptr1
    type(c_ptr) function ptr2(a, n)
      real(c_double), allocatable, target, intent(in) :: a(:,:)
      integer, intent(in) :: n
      ptr2 = c_null_ptr
      if (n > 0) ptr2 = c_loc(a)
    end function ptr2
    type(c_ptr) function ptr3(a, n)
      real(c_double), allocatable, target, intent(in) :: a(:,:,:)
      integer, intent(in) :: n
      ptr3 = c_null_ptr
      if (n > 0) ptr3 = c_loc(a)
    end function ptr3
  end subroutine upload_domain

  !> Do a single PartMC timestep: partmc_timestep (interface/wrf_pmc_driver.F90:46-290) with its
  !> argument list, the loop body of every owned cell as ONE device call (pmcx_timestep): mc_coag on
  !> chemistry steps, scenario_update_gas_state, scenario_update_aero_state, seasalt_emissions,
  !> aero_state_rebalance, over the cells of the reference's loop nest.  MOSAIC (driver:232-245) is
  !> not part of the library: grid%do_mosaic must be off.
  recursive subroutine partmc_timestep_pmcx(grid, config_flags, scenario, env_states, &
       aero_data, aero_states, gas_data, gas_states, pmc_is, pmc_ie, pmc_js, &
       pmc_je, pmc_ks, pmc_ke, global_nx, global_ny, global_nz)
    integer, intent(in) :: pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke
    type(domain), intent(inout) :: grid
    type(grid_config_rec_type), intent(in) :: config_flags
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: aero_states
    type(scenario_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: scenario
    type(env_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: env_states
    type(gas_data_t), intent(in) :: gas_data
    type(gas_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: gas_states
    integer :: global_nx, global_ny, global_nz

    integer :: nx, nz, ny, n_cells, i, j, k, c
    real(kind=dp) :: elapsed_time
    real(c_double), allocatable, target :: old_height(:), height(:), temp(:), pres(:), rh(:), wind(:)
    real(c_double), allocatable, target :: em_rate(:,:), bg_conc(:,:), dil(:)
    integer(c_int32_t), allocatable, target :: ocean(:), st(:), ns(:), nc(:)
    integer(c_int32_t), target :: ss_class(5), ss_source(5)
    type(pmcx_step_flags_t) :: flags
    type(pmcx_aero_dist_t), target :: d_em, d_bg
    logical :: changed
    integer :: m

    ! a rank of a cut domain (pmcx_shim_set_neighbours): the same routine on the patch plus its ghost ring
    if (associated(shift_proc) .and. .not. in_padded) then
       call pad_states(grid, scenario, env_states, aero_states, gas_states, pmc_is, pmc_ie, pmc_js, pmc_je, &
            pmc_ks, pmc_ke, global_nx, global_ny)
       in_padded = .true.
       call partmc_timestep_pmcx(grid, config_flags, p_scn, p_env, aero_data, p_aero, gas_data, p_gas, &
            pmc_is - 1, pmc_ie + 1, pmc_js - 1, pmc_je + 1, pmc_ks, pmc_ke, global_nx, global_ny, global_nz)
       in_padded = .false.
       env_states = p_env(pmc_is:pmc_ie, pmc_ks:pmc_ke, pmc_js:pmc_je)
       return
    end if
    if (grid%do_mosaic) then
       write(0,'(a)') 'partmc_timestep_pmcx: grid%do_mosaic is on; MOSAIC is not part of libpmcx &
            &(run it on host states between partmc_sync_host_pmcx and a new upload, or switch it off)'
       call pmcx_die(4, 'partmc_timestep_pmcx')
    end if
    nx = pmc_ie - pmc_is + 1; nz = pmc_ke - pmc_ks + 1; ny = pmc_je - pmc_js + 1
    n_cells = nx * nz * ny
    allocate(old_height(n_cells), height(n_cells), temp(n_cells), pres(n_cells), rh(n_cells))

    ! Update the meteorology (driver:120-135)
    call wrf_to_partmc_pmcx(grid, nx, nz, ny, pmc_is, pmc_ks, pmc_js, env_states, old_height)
    elapsed_time = real(real_time_since_sim_start(grid), kind=dp)
    if (in_padded .and. .not. (config_flags%periodic_x .and. config_flags%periodic_y)) then
       ! a ghost cell beyond the domain's open edge stands for nothing (no rank owns it, nothing is sent
       ! there): whatever grid%'s halo holds there, it gets the meteorology of the nearest cell of the domain
       do j = pmc_js, pmc_je
       do k = pmc_ks, pmc_ke
       do i = pmc_is, pmc_ie
          if (i >= 1 .and. i <= global_nx .and. j >= 1 .and. j <= global_ny) cycle
          env_states(i,k,j) = env_states(min(max(i, 1), global_nx), k, min(max(j, 1), global_ny))
          old_height(((j - pmc_js) * nz + (k - pmc_ks)) * nx + (i - pmc_is) + 1) = &
               old_height(((min(max(j, 1), global_ny) - pmc_js) * nz + (k - pmc_ks)) * nx + (min(max(i, 1), global_nx) - pmc_is) + 1)
       end do
       end do
       end do
    end if
    c = 0
    do j = pmc_js, pmc_je
    do k = pmc_ks, pmc_ke
    do i = pmc_is, pmc_ie
       c = c + 1
       env_states(i,k,j)%elapsed_time = elapsed_time
       temp(c) = env_states(i,k,j)%temp
       pres(c) = env_states(i,k,j)%pressure
       rh(c) = env_states(i,k,j)%rel_humid
       height(c) = env_states(i,k,j)%height
       if (.not. (old_height(c) > 0d0)) old_height(c) = height(c)   ! (no previous state yet)
    end do
    end do
    end do

    if (.not. resident) then
       ! room for a rebalance's doubling and a step's arrivals: the headroom (default 3) times the
       ! ideal count (the reference's num_particles); pmcx_shim_set_headroom
       call upload_domain(config_flags, scenario, env_states, aero_data, aero_states, gas_data, &
            gas_states, nx, nz, ny, pmc_is, pmc_js, global_nx, global_ny, &
            int(headroom * grid%num_particles))
       resident = .true.
       res_nx = nx; res_nz = nz; res_ny = ny
       if (in_padded) then
          ! the block's place in the domain: a row of it is shorter than a row of the domain, so the
          ! global id of a cell -- the key of every random draw -- is formed from both
          call pmcx_die(pmcx_set_cell_id_base(ctx, u32(int((pmc_js - 1) * nz, 8) * global_nx + (pmc_is - 1))), &
               'pmcx_set_cell_id_base')
          call pmcx_die(pmcx_set_transport_cuts(ctx, 1, nx, pmc_is - 1, global_nx), 'pmcx_set_transport_cuts')
       end if
    else if (nx /= res_nx .or. nz /= res_nz .or. ny /= res_ny) then
       call pmcx_die(4, 'partmc_timestep_pmcx (the resident block has another shape)')
    end if
    call pmcx_die(pmcx_set_env(ctx, temp, pres, rh), 'pmcx_set_env')

    call pmcx_step_flags_default(flags)
    flags%do_coagulation = merge(1, 0, grid%do_coagulation)                     ! driver:190
    flags%allow_doubling = merge(1, 0, grid%allow_doubling)                    ! driver:249
    flags%allow_halving = merge(1, 0, grid%allow_halving)
    flags%kernel_type = COAG_KERNEL_TYPE_BROWN                                 ! driver:192
    flags%dt = real(grid%dt, dp)                                               ! driver:203
    flags%partmc_chem_dt = real(grid%partmc_chem_dt, kind=dp)                  ! driver:155
    flags%elapsed_time = elapsed_time
    flags%random_seed = int(grid%random_seed, c_int64_t)                       ! wrf_pmc_init.F90:131
    flags%height = c_loc(height)
    flags%old_height = c_loc(old_height)

    ! scenario(i,k,j) at this time: the modes live on the device until a cell's series moves on;
    ! what travels every step is one number per (cell, mode) -- the emission rate
    ! num_conc * rate_scale / height [# m^-3 s^-1], the background concentration -- and the cells'
    ! dilution rates
    call scenario_modes(0, n_cells, scenario, elapsed_time, cfg%n_spec, em_n_mode, em_level, changed, &
         em_num, em_gm, em_gs, em_vf, em_cls, em_src, em_scale)
    if (em_n_mode > 0) then
       d_em%n_mode = em_n_mode
       if (changed) then
          d_em%num_conc = c_loc(em_num); d_em%gm_radius = c_loc(em_gm); d_em%gsd_log10 = c_loc(em_gs)
          d_em%vol_frac = c_loc(em_vf); d_em%mode_class = c_loc(em_cls); d_em%mode_source = c_loc(em_src)
          d_em%per_cell = 1
          call pmcx_die(pmcx_set_aero_scenario_modes(ctx, 0, c_loc(d_em)), 'pmcx_set_aero_scenario_modes')
       end if
       allocate(em_rate(em_n_mode, n_cells))
       do c = 1, n_cells
          do m = 1, em_n_mode
             em_rate(m, c) = -1d0                                         ! (a mode the cell does not have)
             if (em_num(m, c) >= 0d0) em_rate(m, c) = em_num(m, c) * (em_scale(c) / height(c))
          end do
       end do
       d_em%num_conc = c_loc(em_rate)
       d_em%gm_radius = c_null_ptr; d_em%gsd_log10 = c_null_ptr; d_em%vol_frac = c_null_ptr
       d_em%mode_class = c_null_ptr; d_em%mode_source = c_null_ptr; d_em%per_cell = 0
       flags%aero_emission = c_loc(d_em)
    end if
    call scenario_modes(1, n_cells, scenario, elapsed_time, cfg%n_spec, bg_n_mode, bg_level, changed, &
         bg_num, bg_gm, bg_gs, bg_vf, bg_cls, bg_src, bg_rate)
    if (bg_n_mode > 0) then
       d_bg%n_mode = bg_n_mode
       if (changed) then
          d_bg%num_conc = c_loc(bg_num); d_bg%gm_radius = c_loc(bg_gm); d_bg%gsd_log10 = c_loc(bg_gs)
          d_bg%vol_frac = c_loc(bg_vf); d_bg%mode_class = c_loc(bg_cls); d_bg%mode_source = c_loc(bg_src)
          d_bg%per_cell = 1
          call pmcx_die(pmcx_set_aero_scenario_modes(ctx, 1, c_loc(d_bg)), 'pmcx_set_aero_scenario_modes')
       end if
       d_bg%num_conc = c_loc(bg_num)
       d_bg%gm_radius = c_null_ptr; d_bg%gsd_log10 = c_null_ptr; d_bg%vol_frac = c_null_ptr
       d_bg%mode_class = c_null_ptr; d_bg%mode_source = c_null_ptr; d_bg%per_cell = 0
       flags%aero_background = c_loc(d_bg)
       flags%aero_dilution_rate = c_loc(bg_rate)
    end if

    ! sea salt on the lowest level of ocean cells (driver:212-223)
    if (grid%partmc_seasalt_param > 0) then
       allocate(ocean(n_cells), wind(n_cells))
       ocean = 0; wind = 0d0
       c = 0
       do j = pmc_js, pmc_je
       do k = pmc_ks, pmc_ke
       do i = pmc_is, pmc_ie
          c = c + 1
          if (k == 1) then
             if (grid%xland(i,j) > 1.5) then
                ocean(c) = 1
                wind(c) = (grid%u10(i,j) * grid%u10(i,j) + grid%v10(i,j) * grid%v10(i,j))**.5
             end if
          end if
       end do
       end do
       end do
       flags%sea_salt_opt = grid%partmc_seasalt_param
       flags%i_na = aero_data_spec_by_name_pmcx(aero_data, 'Na') - 1           ! driver:1828-1829
       flags%i_cl = aero_data_spec_by_name_pmcx(aero_data, 'Cl') - 1
       do m = 1, 5                                                            ! "sea_salt_NN", driver:1841-1846
          ss_class(m) = name_index(aero_data%weight_class_name, 'sea_salt_', m) - 1
          ss_source(m) = name_index(aero_data%source_name, 'sea_salt_', m) - 1
       end do
       flags%ocean_surface = c_loc(ocean); flags%wind_speed = c_loc(wind)
       flags%ss_mode_class = c_loc(ss_class); flags%ss_mode_source = c_loc(ss_source)
    end if

    ! the allow_doubling / allow_halving the reference hands to every add of particles from a
    ! distribution (driver:207-210, 1920-1922; wrf_pmc_trans_aero.F90:1030-1032): the weight of a
    ! class is prepared for the add; it stays in force for the transport driver's boundary refill
    call pmcx_die(pmcx_set_weight_preparation(ctx, 1, merge(1, 0, grid%allow_doubling), &
         merge(1, 0, grid%allow_halving)), 'pmcx_set_weight_preparation')
    call pmcx_die(pmcx_timestep(ctx, flags), 'pmcx_timestep')
    ! the reference aborts inside PartMC; here a cell's status bits say what went wrong
    allocate(ns(n_cells), nc(n_cells), st(n_cells))
    call pmcx_die(pmcx_get_coag_counters(ctx, ns, nc, st), 'pmcx_get_coag_counters')
    if (any(iand(st, 3) /= 0)) call pmcx_die(5, 'partmc_timestep_pmcx (event / append log overflow in mc_coag)')
    if (any(iand(st, 16) /= 0)) call pmcx_die(5, 'partmc_timestep_pmcx (a cell ran out of particle slots)')
    if (any(iand(st, 32) /= 0)) call pmcx_die(5, 'partmc_timestep_pmcx (particle created above the kernel-bounds range)')
    last_n_samp = 0; last_n_coag = 0
    if (mod(elapsed_time, flags%partmc_chem_dt) == 0.0d0 .and. grid%do_coagulation) then   ! driver:189-190
       last_n_samp = sum(int(ns, 8)); last_n_coag = sum(int(nc, 8))
    end if
  end subroutine partmc_timestep_pmcx

  integer function aero_data_spec_by_name_pmcx(aero_data, name)
    type(aero_data_t), intent(in) :: aero_data
    character(len=*), intent(in) :: name
    integer :: s
    aero_data_spec_by_name_pmcx = 0
    if (.not. allocated(aero_data%name)) return
    do s = 1, size(aero_data%name)
       if (trim(aero_data%name(s)) == name) aero_data_spec_by_name_pmcx = s
    end do
  end function aero_data_spec_by_name_pmcx

  integer function name_index(names, prefix, m)
    character(len=*), allocatable, intent(in) :: names(:)
    character(len=*), intent(in) :: prefix
    integer, intent(in) :: m
    character(len=AERO_SOURCE_NAME_LEN) :: want
    integer :: s
    name_index = 1
    if (.not. allocated(names)) return
    write(want, '(a,i2.2)') prefix, m
    do s = 1, size(names)
       if (trim(names(s)) == trim(want)) name_index = s
    end do
  end function name_index

  !> depvel_partmc (interface/wrf_pmc_dep.F90:190-215): aerodynamic resistance from the inverse
  !> Monin-Obukhov length, the roughness length and the friction velocity.  In a WRF build the
  !> reference's own routine is used instead.
  subroutine depvel_pmcx(rmol, zr, z0, ustar, aer_res)
    real, intent(inout) :: rmol
    real(kind=dp), intent(in) :: zr, z0, ustar
    real(kind=dp), intent(out) :: aer_res
    real(kind=dp) :: ao, ar, polint, vk
    vk = 0.40d0
    if (abs(rmol) < 1.0e-6) rmol = 0.
    if (rmol < 0.0) then
       ar = ((1.0 - 9.0 * zr * rmol)**(0.25) + 0.001)**2
       ao = ((1.0 - 9.0 * z0 * rmol)**(0.25) + 0.001)**2
       polint = 0.74 * (dlog((ar - 1.0) / (ar + 1.0)) - dlog((ao - 1.0) / (ao + 1.0)))
    else if (rmol == 0.0) then
       polint = 0.74 * dlog(zr / z0)
    else
       polint = 0.74 * dlog(zr / z0) + 4.7 * rmol * (zr - z0)
    end if
    aer_res = polint / (vk * max(ustar, 1.0e-4))
  end subroutine depvel_pmcx

  !> Top level subroutine for aerosol transport and deposition: wrf_pmc_trans_driver
  !> (interface/wrf_pmc_trans.F90:37-200) with its argument list, on the resident block.  The
  !> probabilities are built on the device from grid%'s fields (compute_advect_probs_wrf,
  !> compute_diffusion_probs, compute_vertical_probs, normalize_probs), then trans_aero:
  !> trans_aero_preweight, the sampling, the rebalance of the interior cells, the refill of the
  !> lateral boundary cells (wrf_pmc_trans_aero.F90:40-971); then dry deposition on the lowest level.
  recursive subroutine wrf_pmc_trans_driver_pmcx(grid, config_flags, env_states, scenario, &
       aero_data, aero_states, gas_states, pmc_is, pmc_ie, pmc_js, pmc_je, &
       pmc_ks, pmc_ke, global_nx, global_ny, global_nz)
    integer, intent(in) :: pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke
    type(domain), intent(inout) :: grid
    type(grid_config_rec_type), intent(in) :: config_flags
    type(env_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: env_states
    type(scenario_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: scenario
    type(aero_data_t), intent(in) :: aero_data
    type(aero_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: aero_states
    type(gas_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: gas_states
    integer, intent(in) :: global_nx, global_ny, global_nz

    integer :: nx, nz, ny, n_cells, n_class, i, j, k, c, step, n_copy, n_res, q, m, p, nm
    logical :: ideal
    real(kind=dp) :: elapsed_time, zr
    type(pmcx_trans_fields_t) :: f
    ! (staging arrays of n_cells and more: page-locked, see pinned())
    real(c_float), pointer :: alt(:,:,:), xkhh(:,:,:), uf(:,:,:,:), vf(:,:,:,:), wf(:,:,:,:), &
         z(:,:,:), zw(:,:,:), exch(:,:,:), f3(:,:,:)
    real(c_double), pointer :: cell_volume(:), expected(:), ra(:), ust(:), alpha_c(:), biga_c(:), height(:), rrho(:)
    real(c_double), allocatable, target :: dens(:), conc(:,:), gm(:,:), gs(:,:), vfr(:,:,:)
    integer(c_int32_t), pointer :: n_moved(:), n_arrived(:), n_rem(:), surface(:), bad(:), n_dep(:), unreal(:)
    integer(c_int32_t), allocatable, target :: from_c(:), to_c(:), res_c(:), &
         thr(:), cls(:,:), src(:,:), n_add(:), n_rm(:), st(:), ns(:), nc(:)
    type(pmcx_aero_dist_t) :: bgd
    integer :: r_is, r_ie, r_js, r_je       ! this rank's cells (the patch inside a padded block)

    if (.not. resident) call pmcx_die(4, 'wrf_pmc_trans_driver_pmcx (partmc_timestep_pmcx comes first: nothing is resident)')
    ! a rank of a cut domain (pmcx_shim_set_neighbours): the same routine on the patch plus its ghost ring
    if (associated(shift_proc) .and. .not. in_padded) then
       in_padded = .true.
       call wrf_pmc_trans_driver_pmcx(grid, config_flags, p_env, p_scn, aero_data, p_aero, p_gas, &
            pmc_is - 1, pmc_ie + 1, pmc_js - 1, pmc_je + 1, pmc_ks, pmc_ke, global_nx, global_ny, global_nz)
       in_padded = .false.
       env_states = p_env(pmc_is:pmc_ie, pmc_ks:pmc_ke, pmc_js:pmc_je)
       return
    end if
    r_is = pmc_is; r_ie = pmc_ie; r_js = pmc_js; r_je = pmc_je
    if (in_padded) then
       r_is = real_is; r_ie = real_ie; r_js = real_js; r_je = real_je
    end if
    nx = pmc_ie - pmc_is + 1; nz = pmc_ke - pmc_ks + 1; ny = pmc_je - pmc_js + 1
    n_cells = nx * nz * ny
    n_class = cfg%n_class
    ideal = config_flags%periodic_x .and. config_flags%periodic_y                  ! trans_aero:127-131
    if (config_flags%periodic_x .neqv. config_flags%periodic_y) &
         call pmcx_die(1, 'wrf_pmc_trans_driver_pmcx (periodic in one direction only: not supported)')
    elapsed_time = env_states(pmc_is,pmc_ks,pmc_js)%elapsed_time
    step = int(elapsed_time / real(grid%dt, dp))            ! the WRF-step count pmcx_timestep uses

    if (grid%do_transport) then
       ! the fields the probabilities read, single precision as grid% holds them (wrf_pmc_trans.F90:
       ! 227-275, 374-406, 451-463, 645-646), copied out of WRF's memory order tile by tile
       call c_f_pointer(pinned(1, 4_c_int64_t * (nx+2) * nz * (ny+2)), f3, [nx+2, nz, ny+2]); alt(0:, 1:, 0:) => f3
       call c_f_pointer(pinned(2, 4_c_int64_t * (nx+2) * nz * (ny+2)), f3, [nx+2, nz, ny+2]); xkhh(0:, 1:, 0:) => f3
       call c_f_pointer(pinned(3, 4_c_int64_t * (nx+1) * nz * ny * n_class), uf, [nx+1, nz, ny, n_class])
       call c_f_pointer(pinned(4, 4_c_int64_t * nx * nz * (ny+1) * n_class), vf, [nx, nz, ny+1, n_class])
       call c_f_pointer(pinned(5, 4_c_int64_t * nx * (nz+1) * ny * n_class), wf, [nx, nz+1, ny, n_class])
       call c_f_pointer(pinned(6, 4_c_int64_t * nx * nz * ny), z, [nx, nz, ny])
       call c_f_pointer(pinned(7, 4_c_int64_t * nx * (nz+1) * ny), zw, [nx, nz+1, ny])
       call c_f_pointer(pinned(8, 4_c_int64_t * nx * (nz+1) * ny), exch, [nx, nz+1, ny])
       alt = grid%alt(pmc_is-1:pmc_ie+1, pmc_ks:pmc_ke, pmc_js-1:pmc_je+1)
       xkhh = grid%xkhh(pmc_is-1:pmc_ie+1, pmc_ks:pmc_ke, pmc_js-1:pmc_je+1)
       uf = grid%u_flux(pmc_is:pmc_ie+1, pmc_ks:pmc_ke, pmc_js:pmc_je, 2:n_class+1)   ! wrf_class = i_class + 1
       vf = grid%v_flux(pmc_is:pmc_ie, pmc_ks:pmc_ke, pmc_js:pmc_je+1, 2:n_class+1)
       wf = grid%w_flux(pmc_is:pmc_ie, pmc_ks:pmc_ke+1, pmc_js:pmc_je, 2:n_class+1)
       z = grid%z(pmc_is:pmc_ie, pmc_ks:pmc_ke, pmc_js:pmc_je)
       zw = grid%z_at_w(pmc_is:pmc_ie, pmc_ks:pmc_ke+1, pmc_js:pmc_je)
       exch = grid%exch_h(pmc_is:pmc_ie, pmc_ks:pmc_ke+1, pmc_js:pmc_je)
       f%nx = nx; f%nz = nz; f%ny = ny
       f%do_vertmix = config_flags%vertmix_onoff                                   ! wrf_pmc_trans.F90:143
       f%all_columns = merge(1, 0, config_flags%do_uniform .or. config_flags%do_scm .or. ideal)   ! :116-136
       f%i_offset = pmc_is - 1; f%j_offset = pmc_js - 1
       f%global_nx = global_nx; f%global_ny = global_ny
       f%dt = real(grid%dt, dp); f%dx = real(grid%dx, dp); f%dy = real(grid%dy, dp)
       f%alt = c_loc(alt); f%xkhh = c_loc(xkhh); f%u_flux = c_loc(uf); f%v_flux = c_loc(vf)
       f%w_flux = c_loc(wf); f%z = c_loc(z); f%z_at_w = c_loc(zw); f%exch_h = c_loc(exch)
       ! (the stage calls below only enqueue, pmcx_set_async: what they report -- normalize_probs'
       ! assertions, the movers, cells out of slots, the deposited -- is looked at once, behind the
       ! pmcx_synchronize at the end of this routine)
       if (.not. in_padded) call pmcx_die(pmcx_set_async(ctx, 1), 'pmcx_set_async')   ! (a cut domain's calls wait for their exchanges)
       call c_f_pointer(pinned(9, 4_c_int64_t * n_cells), bad, [n_cells])
       call pmcx_die(pmcx_transport_probs(ctx, f, c_null_ptr, c_null_ptr, c_null_ptr, c_loc(bad), &
            c_null_ptr), 'pmcx_transport_probs')

       call c_f_pointer(pinned(10, 8_c_int64_t * n_cells), cell_volume, [n_cells])
       call c_f_pointer(pinned(11, 8_c_int64_t * n_cells * n_class), expected, [n_cells * n_class])
       call c_f_pointer(pinned(12, 4_c_int64_t * n_cells), n_rem, [n_cells])
       call c_f_pointer(pinned(13, 4_c_int64_t * n_cells), n_moved, [n_cells])
       call c_f_pointer(pinned(14, 4_c_int64_t * n_cells), n_arrived, [n_cells])
       c = 0
       do j = pmc_js, pmc_je
       do k = pmc_ks, pmc_ke
       do i = pmc_is, pmc_ie
          c = c + 1
          cell_volume(c) = env_states(i,k,j)%cell_volume
       end do
       end do
       end do
       ! trans_aero (wrf_pmc_trans_aero.F90:40-971); the sampling is the context's default law, the one
       ! that has the reference's variance where weights differ (movers arrive once, emitters send
       ! the other copies: pmcx_set_transport_law, :2236-2300), for any number of weight classes
       if (in_padded) then
          call pmcx_die(pmcx_set_transport_lateral(ctx, merge(0, 1, ideal), pmc_js - 1, global_ny), &
               'pmcx_set_transport_lateral')
          call transport_cut()                      ! trans_aero:148-273, 294, 381-473, 519-801, 1261-1343
       else
       call pmcx_die(pmcx_set_transport_lateral(ctx, merge(0, 1, ideal), 0, 0), 'pmcx_set_transport_lateral')
       call pmcx_die(pmcx_transport_preweight(ctx, nx, nz, ny, merge(1, 0, ideal), merge(1, 0, ideal), &
            c_null_ptr, c_null_ptr, cell_volume, int(grid%random_seed, c_int64_t), step, expected, n_rem), &
            'pmcx_transport_preweight')                                           ! trans_aero:294
       call pmcx_die(pmcx_transport(ctx, nx, nz, ny, merge(1, 0, ideal), merge(1, 0, ideal), c_null_ptr, &
            c_null_ptr, cell_volume, int(grid%random_seed, c_int64_t), step, n_moved, n_arrived), &
            'pmcx_transport')                                                     ! trans_aero:519-801
       end if
       ! aero_state_rebalance + aero_state_sort of the interior cells (trans_aero:798-801): its draws
       ! are keyed apart from those of partmc_timestep's rebalance in the same WRF step
       call pmcx_die(pmcx_rebalance(ctx, merge(1, 0, grid%allow_doubling), merge(1, 0, grid%allow_halving), &
            int(grid%random_seed, c_int64_t), ior(step, ishft(1, 30))), 'pmcx_rebalance')

       if (.not. ideal) then
          ! (pmcx_boundary_resample starts a new status word: the transport's is looked at first)
          call pmcx_die(pmcx_synchronize(ctx), 'pmcx_synchronize')
          call check_transport_status('in the transport')
          ! the refill of the lateral boundary cells (trans_aero:803-955): which copy their inner
          ! neighbour (outflow), which are resampled against the background (inflow; emptied first
          ! where the wind has turned or the background has moved to its next time level)
          allocate(from_c(n_cells), to_c(n_cells), res_c(n_cells), thr(n_cells))
          n_copy = 0; n_res = 0
          ! (a rank of a cut domain sees to the boundary cells it holds; a patch on a lateral boundary is
          ! at least two cells wide, as WRF's are: a boundary cell's inner neighbour is this rank's too)
          if (r_is == 1) call side(1, 0, 1, 0, .true.)                            ! west,  u_2(i,k,j) < 0: outflow
          if (r_ie == global_nx) call side(global_nx, 0, -1, 0, .false.)          ! east,  u_2(i+1,k,j) > 0
          if (r_js == 1) call side(0, 1, 0, 1, .true.)                            ! south, v_2(i,k,j) < 0
          if (r_je == global_ny) call side(0, global_ny, 0, -1, .false.)          ! north, v_2(i,k,j+1) > 0
          if (n_copy > 0) call pmcx_die(pmcx_boundary_copy(ctx, n_copy, from_c, to_c), 'pmcx_boundary_copy')
          if (n_res > 0) then
             nm = 0
             do q = 1, n_res
                call cell_ijk(res_c(q), i, k, j)
                nm = max(nm, size(scenario(i,k,j)%aero_background(1)%mode))
             end do
             allocate(conc(nm, n_res), gm(nm, n_res), gs(nm, n_res), vfr(cfg%n_spec, nm, n_res), cls(nm, n_res), &
                  src(nm, n_res), dens(n_res), n_add(n_res), n_rm(n_res))
             conc = -1d0; gm = 1d0; gs = 0.1d0; vfr = 0d0; vfr(1,:,:) = 1d0; cls = 0; src = 0   ! (-1: no such mode)
             do q = 1, n_res
                call cell_ijk(res_c(q), i, k, j)
                p = dist_level(size(scenario(i,k,j)%aero_dilution_time), scenario(i,k,j)%aero_dilution_time, &
                     env_states(i,k,j)%elapsed_time)                              ! aero_dist_interp_1d, trans_aero:834-838
                do m = 1, size(scenario(i,k,j)%aero_background(p)%mode)
                   conc(m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%num_conc
                   gm(m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%char_radius
                   gs(m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%log10_std_dev_radius
                   vfr(:, m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%vol_frac(1:cfg%n_spec)
                   cls(m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%weight_class - 1
                   src(m, q) = scenario(i,k,j)%aero_background(p)%mode(m)%source - 1
                end do
                dens(q) = 1.0d0 / env_states(i,k,j)%inverse_density                ! trans_aero:842
             end do
             bgd%n_mode = nm
             bgd%num_conc = c_loc(conc); bgd%gm_radius = c_loc(gm); bgd%gsd_log10 = c_loc(gs)
             bgd%vol_frac = c_loc(vfr); bgd%mode_class = c_loc(cls); bgd%mode_source = c_loc(src)
             bgd%per_cell = 1
             call pmcx_die(pmcx_boundary_resample(ctx, n_res, res_c, thr, bgd, dens, 0.0d0, &
                  int(grid%random_seed, c_int64_t), step, n_add, n_rm), 'pmcx_boundary_resample')
          end if
       end if
    end if

    ! Do deposition on lowest layer (wrf_pmc_trans.F90:182-198, wrf_pmc_dep.F90:30-96,
    ! wrf_pmc_dep_aero.F90:28-151)
    if (pmc_ks == 1 .and. grid%do_deposition) then
       call c_f_pointer(pinned(15, 4_c_int64_t * n_cells), surface, [n_cells])
       call c_f_pointer(pinned(16, 8_c_int64_t * n_cells), ra, [n_cells])
       call c_f_pointer(pinned(17, 8_c_int64_t * n_cells), ust, [n_cells])
       call c_f_pointer(pinned(18, 8_c_int64_t * n_cells), alpha_c, [n_cells])
       call c_f_pointer(pinned(19, 8_c_int64_t * n_cells), biga_c, [n_cells])
       call c_f_pointer(pinned(20, 8_c_int64_t * n_cells), height, [n_cells])
       call c_f_pointer(pinned(21, 8_c_int64_t * n_cells), rrho, [n_cells])
       call c_f_pointer(pinned(22, 4_c_int64_t * n_cells), n_dep, [n_cells])
       call c_f_pointer(pinned(23, 4_c_int64_t * n_cells), unreal, [n_cells])
       surface = 0; ra = 1d0; ust = 1d0; alpha_c = 1d0; biga_c = 1d0; height = 1d0; rrho = 1d0
       zr = 2.0
       do j = max(r_js, 2), min(r_je, global_ny - 1)
          do i = max(r_is, 2), min(r_ie, global_nx - 1)
             c = ((j - pmc_js) * nz + 0) * nx + (i - pmc_is) + 1
             surface(c) = 1
             ust(c) = real(grid%ust(i,j), dp)
             call depvel_pmcx(grid%rmol(i,j), zr, real(grid%znt(i,j), kind=dp), ust(c), ra(c))
             call zhang_parameters(grid%ivgtyp(i,j), alpha_c(c), biga_c(c))
             height(c) = env_states(i,1,j)%height
             rrho(c) = env_states(i,1,j)%inverse_density
          end do
       end do
       call pmcx_die(pmcx_dry_deposition(ctx, real(grid%dt, dp), int(grid%random_seed, c_int64_t), step, &
            surface, ra, ust, alpha_c, biga_c, height, rrho, n_dep, unreal), 'pmcx_dry_deposition')
       last_aer_res = ra
    end if

    ! the step's queue ends here: wait for it, then look at what it reported
    call pmcx_die(pmcx_synchronize(ctx), 'pmcx_synchronize')
    call pmcx_die(pmcx_set_async(ctx, 0), 'pmcx_set_async')
    if (grid%do_transport) then
       ! normalize_probs' assertions (wrf_pmc_trans.F90:713,746,759,771)
       if (in_padded) call clear_ghost_flags(bad)      ! (a ghost cell's fields are its owner's business)
       if (any(bad /= 0)) call pmcx_die(5, 'wrf_pmc_trans_driver_pmcx (normalize_probs: a probability is negative, above one, or their sum is)')
       last_n_moved = sum(int(n_moved, 8))
       ! (open boundaries: the transport's word was looked at before the refill started a new one;
       ! this is the refill's -- a boundary cell resampled against a background it has no room for)
       if (ideal) then
          call check_transport_status('in the transport')
       else
          call check_transport_status('in the lateral boundary refill')
       end if
    end if
    if (pmc_ks == 1 .and. grid%do_deposition) then
       last_n_deposited = sum(int(n_dep, 8))
       ! the reference's message, not an abort (wrf_pmc_dep_aero.F90:194-196; it prints the extreme
       ! probabilities of a cell, which stay on the device here: the cells are counted instead)
       if (any(unreal /= 0)) print *, "Unrealistic deposition in", count(unreal /= 0), "cells"
    end if

  contains

    !> flags of the block's ghost ring cleared
    subroutine clear_ghost_flags(flag)
      integer(c_int32_t), intent(inout) :: flag(nx, nz, ny)
      flag(1,:,:) = 0; flag(nx,:,:) = 0; flag(:,:,1) = 0; flag(:,:,ny) = 0
    end subroutine clear_ghost_flags

    !> trans_aero for a rank of a cut domain: the block is the patch plus a ghost ring, and what crosses a
    !> cut travels through shift_proc -- the preweight's ready terms of the four boundary strips
    !> (wrf_pmc_trans_aero.F90:1261-1343), the strips' scaled weights for the neighbours' ghost strips
    !> (:381-473), the movers (:543-675).  Sides: 0 south, 1 north, 2 west, 3 east.
    subroutine transport_cut()
      integer, parameter :: i8 = c_int64_t
      real(c_double), allocatable, target :: edge(:), got_rows(:,:), got_cols(:,:), mag(:), strip(:,:), gst(:,:)
      integer(c_int64_t) :: nb_s(0:3), nb_r(0:3), sz_s(0:3), sz_r(0:3)
      integer(c_int64_t), target :: n_out(0:3), n_in(0:3)
      type(c_ptr) :: ps(0:3), pr(0:3), dbuf
      integer(c_int8_t), allocatable, target :: hs0(:), hs1(:), hs2(:), hs3(:), hr0(:), hr1(:), hr2(:), hr3(:)
      integer :: sd, row_n, col_n, ii, jj, kk2, q2
      integer(c_int64_t) :: seed64
      seed64 = int(grid%random_seed, c_int64_t)
      row_n = nx * nz * n_class; col_n = ny * nz * n_class
      ! ---- preweight: the ready terms of the boundary strips, rows first then columns
      allocate(edge(2 * row_n + 2 * col_n), got_rows(row_n, 0:1), got_cols(col_n, 0:1))
      call pmcx_die(pmcx_transport_preweight_begin(ctx, nx, nz, ny, merge(1, 0, ideal), merge(1, 0, ideal), &
           pmc_js - 1, global_ny, c_null_ptr, c_null_ptr, cell_volume, edge), 'pmcx_transport_preweight_begin')
      ps = [c_loc(edge(1)), c_loc(edge(row_n + 1)), c_loc(edge(2 * row_n + 1)), c_loc(edge(2 * row_n + col_n + 1))]
      pr = [c_loc(got_rows(1, 0)), c_loc(got_rows(1, 1)), c_loc(got_cols(1, 0)), c_loc(got_cols(1, 1))]
      nb_s = [int(row_n, i8), int(row_n, i8), int(col_n, i8), int(col_n, i8)] * 8_i8
      call swap4(ps, nb_s, pr, nb_s)
      call pmcx_die(pmcx_transport_preweight_x_terms(ctx, merge(pr(2), c_null_ptr, has_nbr(2)), &
           merge(pr(3), c_null_ptr, has_nbr(3)), merge(1, 0, has_nbr(2) .and. r_is == 1), &
           merge(1, 0, has_nbr(3) .and. r_ie == global_nx)), 'pmcx_transport_preweight_x_terms')
      call pmcx_die(pmcx_transport_preweight_finish(ctx, merge(pr(0), c_null_ptr, has_nbr(0)), &
           merge(pr(1), c_null_ptr, has_nbr(1)), merge(1, 0, has_nbr(0) .and. r_js == 1), &
           merge(1, 0, has_nbr(1) .and. r_je == global_ny), seed64, step, expected, n_rem), &
           'pmcx_transport_preweight_finish')                                     ! trans_aero:294
      ! ---- the ghost strips stand for the neighbours' boundary strips: their weights follow the scaling there
      allocate(mag(n_class * n_cells), strip(max(row_n, col_n), 0:3), gst(max(row_n, col_n), 0:3))
      call pmcx_die(pmcx_get_weights(ctx, mag), 'pmcx_get_weights')
      do sd = 0, 3
         q2 = 0
         do jj = 1, ny
         do kk2 = 1, nz
         do ii = 1, nx
            if ((sd == 0 .and. jj == 2) .or. (sd == 1 .and. jj == ny - 1) .or. &
                 (sd == 2 .and. ii == 2) .or. (sd == 3 .and. ii == nx - 1)) then
               c = ((jj - 1) * nz + (kk2 - 1)) * nx + ii
               strip(q2 + 1 : q2 + n_class, sd) = mag((c - 1) * n_class + 1 : c * n_class)
               q2 = q2 + n_class
            end if
         end do
         end do
         end do
      end do
      ps = [c_loc(strip(1, 0)), c_loc(strip(1, 1)), c_loc(strip(1, 2)), c_loc(strip(1, 3))]
      pr = [c_loc(gst(1, 0)), c_loc(gst(1, 1)), c_loc(gst(1, 2)), c_loc(gst(1, 3))]
      call swap4(ps, nb_s, pr, nb_s)
      do sd = 0, 3
         if (.not. has_nbr(sd)) cycle
         q2 = 0
         do jj = 1, ny
         do kk2 = 1, nz
         do ii = 1, nx
            if ((sd == 0 .and. jj == 1) .or. (sd == 1 .and. jj == ny) .or. &
                 (sd == 2 .and. ii == 1) .or. (sd == 3 .and. ii == nx)) then
               c = ((jj - 1) * nz + (kk2 - 1)) * nx + ii
               mag((c - 1) * n_class + 1 : c * n_class) = gst(q2 + 1 : q2 + n_class, sd)
               q2 = q2 + n_class
            end if
         end do
         end do
         end do
      end do
      call pmcx_die(pmcx_set_weights(ctx, mag), 'pmcx_set_weights')
      ! ---- the sampling: every cell draws its movers, the boundary strips' leave through the four sides
      call pmcx_die(pmcx_transport_begin(ctx, nx, nz, ny, 0, c_null_ptr, c_null_ptr, cell_volume, seed64, step), &
           'pmcx_transport_begin')                                                ! trans_aero:519-675
      do sd = 0, 3
         call pmcx_die(pmcx_transport_export_size(ctx, sd, n_out(sd)), 'pmcx_transport_export_size')
      end do
      allocate(hs0(max(n_out(0), 1_i8)), hs1(max(n_out(1), 1_i8)), hs2(max(n_out(2), 1_i8)), hs3(max(n_out(3), 1_i8)))
      ps = [c_loc(hs0), c_loc(hs1), c_loc(hs2), c_loc(hs3)]
      do sd = 0, 3
         if (n_out(sd) == 0 .or. .not. has_nbr(sd)) cycle
         call pmcx_die(pmcx_device_alloc(ctx, n_out(sd), dbuf), 'pmcx_device_alloc')
         call pmcx_die(pmcx_transport_export(ctx, sd, dbuf), 'pmcx_transport_export')
         call pmcx_die(pmcx_device_copy(ctx, ps(sd), dbuf, n_out(sd), 2), 'pmcx_device_copy')
         call pmcx_die(pmcx_device_free(ctx, dbuf), 'pmcx_device_free')
      end do
      ! the sizes first, then the records
      n_in = 0
      sz_s = 8_i8
      call swap4([c_loc(n_out(0)), c_loc(n_out(1)), c_loc(n_out(2)), c_loc(n_out(3))], sz_s, &
           [c_loc(n_in(0)), c_loc(n_in(1)), c_loc(n_in(2)), c_loc(n_in(3))], sz_s)
      allocate(hr0(max(n_in(0), 1_i8)), hr1(max(n_in(1), 1_i8)), hr2(max(n_in(2), 1_i8)), hr3(max(n_in(3), 1_i8)))
      pr = [c_loc(hr0), c_loc(hr1), c_loc(hr2), c_loc(hr3)]
      nb_r = n_in
      where (.not. has_nbr) nb_r = 0
      call swap4(ps, n_out, pr, nb_r)
      do sd = 0, 3
         if (.not. has_nbr(sd) .or. nb_r(sd) == 0) then
            call pmcx_die(pmcx_transport_import(ctx, sd, c_null_ptr, 0_i8), 'pmcx_transport_import')
         else
            call pmcx_die(pmcx_device_alloc(ctx, nb_r(sd), dbuf), 'pmcx_device_alloc')
            call pmcx_die(pmcx_device_copy(ctx, dbuf, pr(sd), nb_r(sd), 1), 'pmcx_device_copy')
            call pmcx_die(pmcx_transport_import(ctx, sd, dbuf, nb_r(sd)), 'pmcx_transport_import')
            call pmcx_die(pmcx_device_free(ctx, dbuf), 'pmcx_device_free')
         end if
      end do
      call pmcx_die(pmcx_transport_finish(ctx, n_moved, n_arrived), 'pmcx_transport_finish')   ! trans_aero:677-801
      call clear_ghost_flags(n_moved)
    end subroutine transport_cut

    !> the cells' status after the transport: no cell may have run out of slots
    subroutine check_transport_status(where)
      character(len=*), intent(in) :: where
      allocate(ns(n_cells), nc(n_cells), st(n_cells))
      call pmcx_die(pmcx_get_coag_counters(ctx, ns, nc, st), 'pmcx_get_coag_counters')
      if (any(iand(st, 16) /= 0)) call pmcx_die(5, 'wrf_pmc_trans_driver_pmcx (a cell ran out of particle slots ' &
           // trim(where) // ': pmcx_shim_set_headroom)')
      if (any(iand(st, 64) /= 0)) call pmcx_die(5, 'wrf_pmc_trans_driver_pmcx (a source cell sends to more (particle, ' &
           // 'destination) pairs than it has slots ' // trim(where) // ': pmcx_shim_set_headroom)')
      if (any(iand(st, 128) /= 0)) call pmcx_die(5, 'wrf_pmc_trans_driver_pmcx (weights of neighbouring cells differ by ' &
           // 'more than three decades ' // trim(where) // ': an emitter met the guard of 4096 copies)')
      deallocate(ns, nc, st)
    end subroutine check_transport_status

    !> 0-based cell number -> WRF's (i, k, j)
    subroutine cell_ijk(c0, i_, k_, j_)
      integer(c_int32_t), intent(in) :: c0
      integer, intent(out) :: i_, k_, j_
      i_ = pmc_is + mod(c0, nx)
      k_ = pmc_ks + mod(c0 / nx, nz)
      j_ = pmc_js + c0 / (nx * nz)
    end subroutine cell_ijk

    !> One lateral side (trans_aero:803-955).  West/east: i_b is the boundary column, inward = +1 /
    !> -1 in x; south/north: j_b the boundary row.  lower: the velocity that decides is the one on
    !> the cell's own lower face (u_2(i,k,j) / v_2(i,k,j)), outflow when negative; otherwise the
    !> upper face (i + 1 / j + 1), outflow when positive.
    subroutine side(i_b, j_b, di, dj, lower)
      integer, intent(in) :: i_b, j_b, di, dj
      logical, intent(in) :: lower
      integer :: ii, jj, kk, a, b, t, i0, j0, p_t, p_tm1
      logical :: change_bc, outflow, turned
      real :: v2, v1
      if (i_b > 0) then
         a = max(r_js, 2); b = min(r_je, global_ny - 1)
         i0 = i_b; j0 = r_js
      else
         a = max(r_is, 2); b = min(r_ie, global_nx - 1)
         i0 = r_is; j0 = j_b
      end if
      ! the background moves to its next time level in this step (decided at the side's first cell,
      ! as the reference does: trans_aero:811-822)
      change_bc = .false.
      if (allocated(scenario(i0,pmc_ks,j0)%aero_dilution_time)) then
         p_tm1 = find_1d(size(scenario(i0,pmc_ks,j0)%aero_background), scenario(i0,pmc_ks,j0)%aero_dilution_time, &
              env_states(i0,pmc_ks,j0)%elapsed_time - grid%dt)
         p_t = find_1d(size(scenario(i0,pmc_ks,j0)%aero_background), scenario(i0,pmc_ks,j0)%aero_dilution_time, &
              env_states(i0,pmc_ks,j0)%elapsed_time)
         change_bc = p_t > p_tm1 .and. p_t > 1
      end if
      do t = a, b
         do kk = pmc_ks, pmc_ke
            if (i_b > 0) then
               ii = i_b; jj = t
               v2 = grid%u_2(merge(ii, ii + 1, lower), kk, jj); v1 = grid%u_1(merge(ii, ii + 1, lower), kk, jj)
            else
               ii = t; jj = j_b
               v2 = grid%v_2(ii, kk, merge(jj, jj + 1, lower)); v1 = grid%v_1(ii, kk, merge(jj, jj + 1, lower))
            end if
            if (lower) then
               outflow = v2 < 0.0d0
            else
               outflow = v2 > 0.0d0
            end if
            c = ((jj - pmc_js) * nz + (kk - pmc_ks)) * nx + (ii - pmc_is)
            if (outflow) then                            ! aero_state_dup_all_particles(inner, boundary)
               n_copy = n_copy + 1
               to_c(n_copy) = c
               from_c(n_copy) = ((jj + dj - pmc_js) * nz + (kk - pmc_ks)) * nx + (ii + di - pmc_is)
            else                                         ! aero_state_resample against the background
               turned = v2 * v1 < 0.0d0
               n_res = n_res + 1
               res_c(n_res) = c
               thr(n_res) = merge(1, 0, turned .or. change_bc)
            end if
         end do
      end do
    end subroutine side

  end subroutine wrf_pmc_trans_driver_pmcx

  !> The land-use parameters of Zhang et al. (2001) the reference's dep_aero_dry_driver looks up
  !> (interface/wrf_pmc_dep_aero.F90:66-151): USGS vegetation type -> Zhang surface type -> alpha and
  !> the collector radius A of season 1 (the reference's fixed iseason; its gamma is overwritten at
  !> :325 and not an input).
  subroutine zhang_parameters(ivgtyp, alpha, big_a)
    integer, intent(in) :: ivgtyp
    real(kind=dp), intent(out) :: alpha, big_a
    integer, parameter :: zhang_from_usgs(25) = (/ 15, 7, 7, 7, 7, 10, 6, 6, 6, 6, 4, 3, 2, 1, 5, 14, 11, &
         10, 8, 9, 10, 9, 8, 12, 8 /)
    real(kind=dp), parameter :: alpha_t(15) = (/ 1.0, 0.6, 1.1, 0.8, 0.8, 1.2, 1.2, 50.0, 50.0, 1.3, 2.0, &
         50.0, 100.0, 100.0, 1.5 /)
    real(kind=dp), parameter :: a_season1(15) = (/ 0.002, 0.005, 0.002, 0.005, 0.005, 0.002, 0.002, 9.999, &
         9.999, 0.010, 0.010, 9.999, 9.999, 9.999, 0.010 /)
    integer :: isurftype
    isurftype = 7
    if (ivgtyp >= 1 .and. ivgtyp <= 25) isurftype = zhang_from_usgs(ivgtyp)
    alpha = alpha_t(isurftype)
    big_a = a_season1(isurftype)
  end subroutine zhang_parameters

  !> Brings the host's aero_states / gas_states up to date with the resident block (what
  !> partmc_before_solve_io needs at output and restart times, wrf_pmc_driver.F90:300-386).
  recursive subroutine partmc_sync_host_pmcx(aero_states, gas_states, pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke)
    integer, intent(in) :: pmc_is, pmc_ie, pmc_js, pmc_je, pmc_ks, pmc_ke
    type(aero_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: aero_states
    type(gas_state_t), dimension(pmc_is:pmc_ie,pmc_ks:pmc_ke,pmc_js:pmc_je), intent(inout) :: gas_states
    integer :: n_cells, n_gas, c, i, j, k
    real(c_double), allocatable :: mix(:,:)
    if (.not. resident) return
    ! a rank of a cut domain: the block is the patch plus its ghost ring (which holds nothing between steps)
    if (associated(shift_proc) .and. .not. in_padded) then
       in_padded = .true.
       call partmc_sync_host_pmcx(p_aero, p_gas, pmc_is - 1, pmc_ie + 1, pmc_js - 1, pmc_je + 1, pmc_ks, pmc_ke)
       in_padded = .false.
       do j = pmc_js, pmc_je
       do k = pmc_ks, pmc_ke
       do i = pmc_is, pmc_ie
          aero_states(i,k,j) = p_aero(i,k,j)
          gas_states(i,k,j) = p_gas(i,k,j)
       end do
       end do
       end do
       return
    end if
    n_cells = (pmc_ie - pmc_is + 1) * (pmc_ke - pmc_ks + 1) * (pmc_je - pmc_js + 1)
    if (n_cells /= cfg%n_cells) call pmcx_die(1, 'partmc_sync_host_pmcx (array shape /= resident block)')
    call download_states_n(n_cells, aero_states)
    if (.not. have_gas) return
    n_gas = size(gas_states(pmc_is,pmc_ks,pmc_js)%mix_rat)
    allocate(mix(n_gas, n_cells))
    call pmcx_die(pmcx_get_gas_state(ctx, mix), 'pmcx_get_gas_state')
    c = 0
    do j = pmc_js, pmc_je
    do k = pmc_ks, pmc_ke
    do i = pmc_is, pmc_ie
       c = c + 1
       gas_states(i,k,j)%mix_rat(1:n_gas) = mix(:, c)
    end do
    end do
    end do
  end subroutine partmc_sync_host_pmcx

  !> The core of partmc_process (interface/wrf_pmc_driver.F90:783,838-857,858-932,1137,1153-1157) on
  !> the resident block: per cell tot_num_conc, tot_mass_conc and the 100-bin num_dist / mass_dist;
  !> optionally the species masses, number by source, PM1/2.5/10 and particle counts.  Arrays are
  !> (bins, cells) / (species, cells) in the cells' memory order.
  subroutine partmc_process_pmcx(tot_num_conc, tot_mass_conc, num_dist, mass_dist, species_mass_conc, &
       num_conc_by_source, pm_mass_conc, n_parts)
    real(kind=dp), intent(out), target :: tot_num_conc(:), tot_mass_conc(:), num_dist(:,:), mass_dist(:,:)
    real(kind=dp), intent(out), target, optional :: species_mass_conc(:,:), num_conc_by_source(:,:), pm_mass_conc(:,:)
    integer(c_int32_t), intent(out), target, optional :: n_parts(:)
    type(c_ptr) :: p1, p2, p3, p4
    if (.not. have_ctx) call pmcx_die(4, 'partmc_process_pmcx (no resident block)')
    ! a rank of a cut domain: arrays over the patch's cells are filled from the block's (patch + ghost ring)
    if (associated(shift_proc) .and. size(tot_num_conc) /= cfg%n_cells .and. &
         size(tot_num_conc) == (real_ie - real_is + 1) * pad_nz * (real_je - real_js + 1)) then
       block
         real(kind=dp), allocatable, target :: tn(:), tm(:), nd(:,:), md(:,:), sm(:,:), sn(:,:), pm(:,:)
         integer(c_int32_t), allocatable, target :: np(:)
         integer :: nb, c, i, j, k, q, ns_src
         nb = cfg%n_cells
         ns_src = max(cfg%n_source, 1)
         allocate(tn(nb), tm(nb), nd(cfg%n_dbin, nb), md(cfg%n_dbin, nb), sm(cfg%n_spec, nb), sn(ns_src, nb), &
              pm(3, nb), np(nb))
         call pmcx_die(pmcx_diagnostics(ctx, c_loc(tn), c_loc(tm), c_loc(nd), c_loc(md)), 'pmcx_diagnostics')
         call pmcx_die(pmcx_diagnostics_ext(ctx, c_loc(sm), c_loc(sn), c_loc(pm), c_loc(np)), 'pmcx_diagnostics_ext')
         q = 0
         do j = real_js, real_je
         do k = 1, pad_nz
         do i = real_is, real_ie
            q = q + 1
            c = ((j - pad_js0) * pad_nz + (k - 1)) * pad_nxb + (i - pad_is0) + 1
            tot_num_conc(q) = tn(c); tot_mass_conc(q) = tm(c)
            num_dist(:, q) = nd(:, c); mass_dist(:, q) = md(:, c)
            if (present(species_mass_conc)) species_mass_conc(:, q) = sm(:, c)
            if (present(num_conc_by_source)) num_conc_by_source(:, q) = sn(1:size(num_conc_by_source, 1), c)
            if (present(pm_mass_conc)) pm_mass_conc(:, q) = pm(:, c)
            if (present(n_parts)) n_parts(q) = np(c)
         end do
         end do
         end do
       end block
       return
    end if
    if (size(tot_num_conc) /= cfg%n_cells .or. size(num_dist, 1) /= cfg%n_dbin) &
         call pmcx_die(1, 'partmc_process_pmcx (array shapes)')
    call pmcx_die(pmcx_diagnostics(ctx, c_loc(tot_num_conc), c_loc(tot_mass_conc), c_loc(num_dist), &
         c_loc(mass_dist)), 'pmcx_diagnostics')
    p1 = c_null_ptr; p2 = c_null_ptr; p3 = c_null_ptr; p4 = c_null_ptr
    if (present(species_mass_conc)) p1 = c_loc(species_mass_conc)
    if (present(num_conc_by_source)) p2 = c_loc(num_conc_by_source)
    if (present(pm_mass_conc)) p3 = c_loc(pm_mass_conc)
    if (present(n_parts)) p4 = c_loc(n_parts)
    if (c_associated(p1) .or. c_associated(p2) .or. c_associated(p3) .or. c_associated(p4)) &
         call pmcx_die(pmcx_diagnostics_ext(ctx, p1, p2, p3, p4), 'pmcx_diagnostics_ext')
  end subroutine partmc_process_pmcx

end module pmcx_shim
