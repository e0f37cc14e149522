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
!   aero_particle_t      wrf_pmc_driver.F90:1419-1466
!   aero_weight_t/array  wrf_pmc_driver.F90:1617-1630
!   apa%n_part           wrf_pmc_trans_aero.F90:2302
!   env_state_t          wrf_pmc_driver.F90:611-637
!   aero_data_t          wrf_pmc_driver.F90:1420
module pmc_mock
  use iso_c_binding
  implicit none

  integer, parameter :: dp = c_double

  integer, parameter :: COAG_KERNEL_TYPE_SEDI = 1
  integer, parameter :: COAG_KERNEL_TYPE_ADDITIVE = 2
  integer, parameter :: COAG_KERNEL_TYPE_CONSTANT = 3
  integer, parameter :: COAG_KERNEL_TYPE_BROWN = 4
  integer, parameter :: COAG_KERNEL_TYPE_ZERO = 5
  integer, parameter :: AERO_WEIGHT_TYPE_FLAT = 2

  type aero_particle_t
     real(kind=dp), allocatable :: vol(:)          ! species volumes (m^3)
     integer, allocatable :: n_orig_part(:)        ! constituents per source
     integer :: weight_group = 1
     integer :: weight_class = 1
     integer(kind=8) :: id = 0
     real(kind=dp) :: least_create_time = 0d0
     real(kind=dp) :: greatest_create_time = 0d0
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

  type env_state_t
     real(kind=dp) :: temp = 288.15d0
     real(kind=dp) :: pressure = 101325d0
     real(kind=dp) :: rel_humid = 0.5d0
     real(kind=dp) :: elapsed_time = 0d0
     integer :: cell_ix = 1, cell_iy = 1, cell_iz = 1
  end type env_state_t

  type aero_data_t
     real(kind=dp), allocatable :: density(:)
     real(kind=dp), allocatable :: kappa(:)
     real(kind=dp), allocatable :: molec_weight(:)
     integer :: i_water = 0
  end type aero_data_t

contains

  integer function aero_state_total_particles(aero_state)
    type(aero_state_t), intent(in) :: aero_state
    aero_state_total_particles = aero_state%apa%n_part
  end function aero_state_total_particles

  integer function aero_data_n_spec(aero_data)
    type(aero_data_t), intent(in) :: aero_data
    aero_data_n_spec = size(aero_data%density)
  end function aero_data_n_spec

end module pmc_mock
