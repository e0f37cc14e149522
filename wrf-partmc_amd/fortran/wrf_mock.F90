! wrf_mock.F90 -- stand-in for the WRF modules the WRF-PartMC interface uses (module_domain,
! module_configure, module_model_constants), declaring ONLY what partmc_timestep and
! wrf_pmc_trans_driver read of `grid` and `config_flags`, with WRF's names, so that
! partmc_timestep_pmcx / wrf_pmc_trans_driver_pmcx (pmcx_shim.F90) can be compiled and exercised
! without WRF.  Inside WRF this module is dropped and the shim uses the real ones (INTEGRATION.md).
!
! Field sources (all under /root/reference/interface):
!   partmc_timestep        wrf_pmc_driver.F90:112,126-127,155-157,190,203,207-220,233,248-250
!   wrf_to_partmc          wrf_pmc_driver.F90:617-634 (p, pb, t_2, moist(:,:,:,p_qv), z_at_w, z, alt)
!   get_grid_cell_volume   wrf_pmc_driver.F90:2014-2016 (rdx, rdy, msftx, msfty, z_at_w)
!   wrf_pmc_trans_driver   wrf_pmc_trans.F90:88,116-131,143,184 (do_transport, do_uniform, do_scm,
!                          periodic_x/y, vertmix_onoff, do_deposition)
!   probabilities          wrf_pmc_trans.F90:227-275 (alt, u_flux, v_flux, w_flux), :374-406 (dx, dy,
!                          dt, xkhh, alt), :451-463,645-646 (z, z_at_w, exch_h)
!   lateral boundaries     wrf_pmc_trans_aero.F90:826-845 ... (u_2, u_1, v_2, v_1)
!   dry deposition         wrf_pmc_dep.F90:77-90 (z_at_w, ust, rmol, znt), wrf_pmc_dep_aero.F90:132
!                          (ivgtyp)
! Arrays are in WRF's memory order (i, k, j) with one halo cell in x and y and one extra level, as
! the interface indexes them.
module wrf_mock
  use iso_c_binding
  implicit none

  integer, parameter :: wdp = c_double
  integer, parameter :: p_qv = 1                        ! module_configure
  real, parameter :: t0 = 300.0, p0 = 100000.0          ! module_model_constants
  real, parameter :: r_d = 287.0, cp = 7.0 * r_d / 2.0, rcp = r_d / cp
  integer, parameter :: PMC_UUID_LEN = 36

  type domain
     ! namelist / registry.partmc (WRFV3/Registry/registry.partmc:1-38)
     real :: dt = 20.0, dx = 1000.0, dy = 1000.0, rdx = 1.0e-3, rdy = 1.0e-3
     real :: partmc_chem_dt = 60.0, num_particles = 1000.0
     integer :: random_seed = 0, partmc_seasalt_param = 0, itimestep = 0
     logical :: do_coagulation = .true., do_mosaic = .false., do_optical = .false.
     logical :: do_transport = .true., do_deposition = .true.
     logical :: allow_doubling = .true., allow_halving = .true.
     character(len=PMC_UUID_LEN) :: uuid = ''
     ! the clock: seconds since the start of the simulation (domain_get_time_since_sim_start)
     real(kind=wdp) :: time_since_sim_start = 0d0
     ! 3-D state (ims:ime, kms:kme, jms:jme)
     real, allocatable :: p(:,:,:), pb(:,:,:), t_2(:,:,:), alt(:,:,:), z(:,:,:), z_at_w(:,:,:)
     real, allocatable :: moist(:,:,:,:)
     real, allocatable :: xkhh(:,:,:), exch_h(:,:,:)
     real, allocatable :: u_2(:,:,:), v_2(:,:,:), u_1(:,:,:), v_1(:,:,:)
     ! class-resolved mass fluxes of the advected number concentrations (registry.partmc_trans):
     ! index 1 is WRF's unused first scalar, class c is c + 1 (wrf_pmc_trans.F90:229)
     real, allocatable :: u_flux(:,:,:,:), v_flux(:,:,:,:), w_flux(:,:,:,:)
     ! 2-D state (ims:ime, jms:jme)
     real, allocatable :: xland(:,:), u10(:,:), v10(:,:), msftx(:,:), msfty(:,:)
     real, allocatable :: ust(:,:), rmol(:,:), znt(:,:)
     integer, allocatable :: ivgtyp(:,:)
  end type domain

  type grid_config_rec_type
     logical :: periodic_x = .false., periodic_y = .false.
     logical :: do_uniform = .false., do_scm = .false.
     integer :: vertmix_onoff = 1
  end type grid_config_rec_type

contains

  !> Allocate the fields of a grid of nx x nz x ny cells (halo cells in x and y -- one unless `halo` says
  !> more: a rank of a cut domain reads two beyond its patch --, nz + 1 levels, n_class + 1 flux scalars).
  subroutine wrf_mock_allocate(grid, nx, nz, ny, n_class, halo)
    type(domain), intent(inout) :: grid
    integer, intent(in) :: nx, nz, ny, n_class
    integer, intent(in), optional :: halo
    integer :: l, hx, hy, hx2, hy2
    l = 0
    if (present(halo)) l = 1 - max(halo, 1)
    hx = nx + 1 - l; hy = ny + 1 - l; hx2 = hx + 1; hy2 = hy + 1
    allocate(grid%p(l:hx, nz+1, l:hy), grid%pb(l:hx, nz+1, l:hy), grid%t_2(l:hx, nz+1, l:hy), &
         grid%alt(l:hx, nz+1, l:hy), grid%z(l:hx, nz+1, l:hy), grid%z_at_w(l:hx, nz+1, l:hy), &
         grid%moist(l:hx, nz+1, l:hy, 1), grid%xkhh(l:hx, nz+1, l:hy), &
         grid%exch_h(l:hx, nz+1, l:hy), grid%u_2(l:hx2, nz+1, l:hy2), grid%v_2(l:hx2, nz+1, l:hy2), &
         grid%u_1(l:hx2, nz+1, l:hy2), grid%v_1(l:hx2, nz+1, l:hy2))
     allocate(grid%u_flux(l:hx2, nz+1, l:hy2, n_class + 1), grid%v_flux(l:hx2, nz+1, l:hy2, n_class + 1), &
         grid%w_flux(l:hx2, nz+1, l:hy2, n_class + 1))
     allocate(grid%xland(l:hx, l:hy), grid%u10(l:hx, l:hy), grid%v10(l:hx, l:hy), &
         grid%msftx(l:hx, l:hy), grid%msfty(l:hx, l:hy), grid%ust(l:hx, l:hy), &
         grid%rmol(l:hx, l:hy), grid%znt(l:hx, l:hy), grid%ivgtyp(l:hx, l:hy))
     grid%p = 0.0; grid%pb = 1.0e5; grid%t_2 = 0.0; grid%alt = 0.85; grid%z = 0.0; grid%z_at_w = 0.0
     grid%moist = 0.0; grid%xkhh = 0.0; grid%exch_h = 0.0
     grid%u_2 = 0.0; grid%v_2 = 0.0; grid%u_1 = 0.0; grid%v_1 = 0.0
     grid%u_flux = 0.0; grid%v_flux = 0.0; grid%w_flux = 0.0
     grid%xland = 1.0; grid%u10 = 0.0; grid%v10 = 0.0; grid%msftx = 1.0; grid%msfty = 1.0
     grid%ust = 0.3; grid%rmol = 0.0; grid%znt = 0.1; grid%ivgtyp = 7
  end subroutine wrf_mock_allocate

  !> module_domain's domain_get_time_since_sim_start followed by real_time, as
  !> wrf_pmc_driver.F90:126-127 combines them: seconds since the start, single precision.
  real function real_time_since_sim_start(grid)
    type(domain), intent(in) :: grid
    real_time_since_sim_start = real(grid%time_since_sim_start)
  end function real_time_since_sim_start

end module wrf_mock
