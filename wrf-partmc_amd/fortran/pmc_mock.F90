! pmc_mock.F90 -- stand-in for the PartMC modules the WRF-PartMC interface uses.
!
! The reference's types live in the un-vendored submodule compdyn/partmc
! (pmc_aero_state, pmc_aero_data, pmc_env_state, ...; .gitmodules:1-3).  This
! module declares ONLY the fields interface/*.F90 touches on the path, with
! the reference's names, so the shim (pmcx_shim.F90) can be compiled and
! exercised without WRF.  When wiring into WRF this module is dropped and
! pmcx_shim uses the real PartMC modules instead (see INTEGRATION.md).
!
! Field sources (all under /root/reference/interface):
!   aero_particle_t      wrf_pmc_driver.F90:1419-1466 (vol, component(:)%source_id/%create_time,
!                        weight_group/class, water_hyst_leg, id, least/greatest_create_time,
!                        absorb/scatter_cross_sect, asymmetry, refract_shell/core, core_vol);
!                        reader wrf_pmc_init.F90:1226-1277
!   aero_data%source_name, %weight_class_name   wrf_pmc_driver.F90:1842-1845
!   aero_weight_t/array  wrf_pmc_driver.F90:1617-1630
!   apa%n_part           wrf_pmc_trans_aero.F90:2302
!   env_state_t          wrf_pmc_driver.F90:611-637
!   aero_data_t          wrf_pmc_driver.F90:1420
!   aero_mode_t, aero_dist_t, scenario_t   wrf_pmc_init.F90:641-657, 747-792, 894-945 (the fields
!                        init_read_in_emissions / init_read_in_bcs fill); consumers
!                        wrf_pmc_driver.F90:202-210, wrf_pmc_trans_aero.F90:811-845
!   gas_state_t, gas_data_t   wrf_pmc_driver.F90:1698, wrf_pmc_init.F90:789-791
module pmc_mock
  use iso_c_binding
  implicit none

  integer, parameter :: dp = c_double
  integer, parameter :: dc = c_double_complex
  integer, parameter :: AERO_SOURCE_NAME_LEN = 100   ! wrf_pmc_init.F90:623
  integer, parameter :: n_swbands = 5                ! wrf_pmc_driver.F90:1405

  integer, parameter :: COAG_KERNEL_TYPE_SEDI = 1
  integer, parameter :: COAG_KERNEL_TYPE_ADDITIVE = 2
  integer, parameter :: COAG_KERNEL_TYPE_CONSTANT = 3
  integer, parameter :: COAG_KERNEL_TYPE_BROWN = 4
  integer, parameter :: COAG_KERNEL_TYPE_ZERO = 5
  integer, parameter :: AERO_WEIGHT_TYPE_FLAT = 2
  integer, parameter :: AERO_MODE_NAME_LEN = 300
  integer, parameter :: AERO_MODE_TYPE_LOG_NORMAL = 1    ! wrf_pmc_init.F90:767

  type aero_component_t                            ! driver:1432-1436
     integer :: source_id = 0
     real(kind=dp) :: create_time = 0d0
  end type aero_component_t

  type aero_particle_t
     real(kind=dp), allocatable :: vol(:)          ! species volumes (m^3), driver:1420
     type(aero_component_t), allocatable :: component(:)   ! one per constituent, driver:1421-1437
     integer :: weight_group = 1                   ! driver:1438-1439
     integer :: weight_class = 1                   ! driver:1440-1441
     integer :: water_hyst_leg = 0                 ! driver:1442-1443
     integer(kind=8) :: id = 0                     ! driver:1447
     real(kind=dp) :: least_create_time = 0d0      ! driver:1448-1449
     real(kind=dp) :: greatest_create_time = 0d0   ! driver:1450-1451
     ! optics, written by MOSAIC when do_optical (driver:1452-1466)
     real(kind=dp), allocatable :: absorb_cross_sect(:), scatter_cross_sect(:), asymmetry(:)
     complex(kind=dc), allocatable :: refract_shell(:), refract_core(:)
     real(kind=dp) :: core_vol = 0d0
  end type aero_particle_t

  type aero_particle_array_t
     integer :: n_part = 0
     type(aero_particle_t), allocatable :: particle(:)
  end type aero_particle_array_t

  type aero_weight_t
     integer :: type = AERO_WEIGHT_TYPE_FLAT
     real(kind=dp) :: magnitude = 0d0
     real(kind=dp) :: exponent = 0d0
  end type aero_weight_t

  type aero_weight_array_t
     type(aero_weight_t), allocatable :: weight(:,:)   ! (group, class)
  end type aero_weight_array_t

  type aero_state_t
     type(aero_particle_array_t) :: apa
     type(aero_weight_array_t) :: awa
     real(kind=dp), allocatable :: n_part_ideal(:,:)   ! (group, class)
     logical :: valid_sort = .false.
  end type aero_state_t

  type env_state_t                                  ! producer wrf_pmc_driver.F90:611-637
     real(kind=dp) :: temp = 288.15d0
     real(kind=dp) :: pressure = 101325d0
     real(kind=dp) :: rel_humid = 0.5d0
     real(kind=dp) :: elapsed_time = 0d0
     real(kind=dp) :: height = 0d0                  ! z_at_w(k+1) - z_at_w(k), driver:627
     real(kind=dp) :: altitude = 0d0, z_min = 0d0, z_max = 0d0
     real(kind=dp) :: inverse_density = 0d0         ! grid%alt, driver:633
     real(kind=dp) :: cell_volume = 0d0             ! get_grid_cell_volume, driver:634
     integer :: cell_ix = 1, cell_iy = 1, cell_iz = 1
  end type env_state_t

  type aero_mode_t                                  ! wrf_pmc_init.F90:759-786
     character(len=AERO_MODE_NAME_LEN) :: name = ''
     integer :: type = AERO_MODE_TYPE_LOG_NORMAL
     real(kind=dp) :: char_radius = 0d0             ! geometric mean radius (m)
     real(kind=dp) :: log10_std_dev_radius = 0d0
     real(kind=dp) :: num_conc = 0d0
     real(kind=dp), allocatable :: vol_frac(:), vol_frac_std(:)
     integer :: source = 1
     integer :: weight_class = 1
  end type aero_mode_t

  type aero_dist_t
     type(aero_mode_t), allocatable :: mode(:)
  end type aero_dist_t

  type gas_state_t                                  ! driver:1698
     real(kind=dp), allocatable :: mix_rat(:)       ! ppb
  end type gas_state_t

  type gas_data_t
     character(len=100), allocatable :: name(:)
  end type gas_data_t

  type scenario_t                                   ! wrf_pmc_init.F90:641-657, 895-915
     real(kind=dp), allocatable :: aero_emission_time(:), aero_emission_rate_scale(:)
     type(aero_dist_t), allocatable :: aero_emission(:)
     real(kind=dp), allocatable :: gas_emission_time(:), gas_emission_rate_scale(:)
     type(gas_state_t), allocatable :: gas_emission(:)
     real(kind=dp), allocatable :: aero_dilution_time(:), aero_dilution_rate(:)
     type(aero_dist_t), allocatable :: aero_background(:)
     real(kind=dp), allocatable :: gas_dilution_time(:), gas_dilution_rate(:)
     type(gas_state_t), allocatable :: gas_background(:)
  end type scenario_t

  type aero_data_t
     character(len=AERO_SOURCE_NAME_LEN), allocatable :: name(:)              ! species names, driver:1828
     real(kind=dp), allocatable :: density(:)
     real(kind=dp), allocatable :: kappa(:)
     real(kind=dp), allocatable :: molec_weight(:)
     integer :: i_water = 0
     character(len=AERO_SOURCE_NAME_LEN), allocatable :: source_name(:)        ! driver:1842
     character(len=AERO_SOURCE_NAME_LEN), allocatable :: weight_class_name(:)  ! driver:1844
  end type aero_data_t

  ! PartMC's particle id counter (module variable next_id, wrf_pmc_driver.F90:1652)
  integer(kind=8), save :: next_id = 1

contains

  integer function aero_state_total_particles(aero_state)
    type(aero_state_t), intent(in) :: aero_state
    aero_state_total_particles = aero_state%apa%n_part
  end function aero_state_total_particles

  integer function aero_data_n_spec(aero_data)
    type(aero_data_t), intent(in) :: aero_data
    aero_data_n_spec = size(aero_data%density)
  end function aero_data_n_spec

  integer function aero_data_n_source(aero_data)         ! driver:862
    type(aero_data_t), intent(in) :: aero_data
    aero_data_n_source = 0
    if (allocated(aero_data%source_name)) aero_data_n_source = size(aero_data%source_name)
  end function aero_data_n_source

  integer function aero_particle_n_components(aero_particle)   ! driver:1422
    type(aero_particle_t), intent(in) :: aero_particle
    aero_particle_n_components = 0
    if (allocated(aero_particle%component)) aero_particle_n_components = size(aero_particle%component)
  end function aero_particle_n_components

  integer function aero_weight_array_n_class(awa)            ! wrf_pmc_trans.F90:142
    type(aero_weight_array_t), intent(in) :: awa
    aero_weight_array_n_class = size(awa%weight, 2)
  end function aero_weight_array_n_class

  integer function gas_data_n_spec(gas_data)                 ! wrf_pmc_init.F90:790
    type(gas_data_t), intent(in) :: gas_data
    gas_data_n_spec = 0
    if (allocated(gas_data%name)) gas_data_n_spec = size(gas_data%name)
  end function gas_data_n_spec

  !> PartMC's find_1d [upstream, util.F90; used in-tree at wrf_pmc_trans_aero.F90:811-816]: the
  !> position p with x_vals(p) <= x < x_vals(p+1); 0 below the first, n from the last on.
  integer function find_1d(n, x_vals, x)
    integer, intent(in) :: n
    real(kind=dp), intent(in) :: x_vals(n), x
    integer :: p
    if (n == 0) then
       find_1d = 0
       return
    end if
    p = 1
    do while (x >= x_vals(p))
       p = p + 1
       if (p > n) exit
    end do
    find_1d = p - 1
  end function find_1d

end module pmc_mock
