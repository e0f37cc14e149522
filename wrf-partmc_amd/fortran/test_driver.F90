! test_driver.F90 -- drives the two drop-in drivers the way WRFV3/frame/module_integrate.F:409-421
! drives the reference: per WRF step
!     call partmc_timestep_pmcx(grid, config_flags, scenario, env_states, aero_data, aero_states, &
!          gas_data, gas_states, pmc_is, ..., global_nz)
!     call wrf_pmc_trans_driver_pmcx(grid, config_flags, env_states, scenario, aero_data, aero_states, &
!          gas_states, pmc_is, ..., global_nz)
! and nothing else of the shim between the first step and the final partmc_sync_host_pmcx.  Input
! and output are raw binary files so that a test can hold the result to the CPU oracle running the
! same loop, bit for bit.
!
! usage: test_driver <state.bin> <fields.bin> <out.bin>
! state.bin : int32 nx, nz, ny, n_part, n_spec, n_class, n_source, n_gas, n_steps, periodic,
!             sea_salt_opt, i_na, i_cl (1-based), n_time_e, n_mode_e, n_time_b, n_mode_b, vertmix;
!             int64 seed; float64 dt, chem_dt, dx, dy, num_particles; float64 density(n_spec);
!             per cell in memory order (i fastest, then k, then j):
!               float64 mag(n_class), n_part_ideal(n_class); the particles as test_shim reads them;
!               float64 gas mix_rat(n_gas);
!               int32 has_emission; if set: float64 time(n_time_e), rate_scale(n_time_e), per time:
!                 per mode float64 num_conc, char_radius, log10_std, vol_frac(n_spec), int32 source,
!                 weight_class (1-based); float64 gas_emission(n_gas)
!               float64 dilution time(n_time_b), aero_dilution_rate(n_time_b), gas_dilution_rate(n_time_b),
!                 per time: per mode as above; float64 gas_background(n_gas)
! fields.bin: per step float64 time_since_sim_start, then grid%'s arrays in the order of
!             read_fields below (single precision, WRF's memory order with halo)
! out.bin   : per step float64 temp, pressure, rel_humid, height, cell_volume, inverse_density
!             (n_cells each), aer_res (n_cells), int32 n_parts(n_cells) after the two calls,
!             int64 n_samp, n_coag, n_moved, n_deposited;
!             then per cell: float64 mag(n_class), gas(n_gas), int32 n_part, the particles as
!             test_shim writes them
program test_driver
  use iso_c_binding
  use pmc_mock
  use wrf_mock
  use pmcx_shim
  implicit none
  character(len=512) :: fstate, ffields, fout
  integer :: nx, nz, ny, n_part, n_spec, n_class, n_source, n_gas, n_steps, periodic, sea_salt_opt, &
       i_na, i_cl, n_time_e, n_mode_e, n_time_b, n_mode_b, vertmix
  integer(kind=8) :: seed
  real(kind=dp) :: dt, chem_dt, dx, dy, num_particles
  type(domain) :: grid
  type(grid_config_rec_type) :: config_flags
  type(aero_data_t) :: aero_data
  type(gas_data_t) :: gas_data
  type(env_state_t), allocatable :: env_states(:,:,:)
  type(aero_state_t), allocatable :: aero_states(:,:,:)
  type(scenario_t), allocatable :: scenario(:,:,:)
  type(gas_state_t), allocatable :: gas_states(:,:,:)
  integer :: u, uf, uo, i, j, k, s, q, m, t, step, has_em, n_cells, n_comp, kk
  integer, allocatable :: cnt(:)
  integer(c_int32_t), allocatable :: n_parts(:)
  real(kind=dp), allocatable :: tn(:), tm(:), nd(:,:), md(:,:)
  real(kind=dp) :: t_lo, t_hi

  call get_command_argument(1, fstate)
  call get_command_argument(2, ffields)
  call get_command_argument(3, fout)
  open(newunit=u, file=trim(fstate), access='stream', form='unformatted', status='old')
  read(u) nx, nz, ny, n_part, n_spec, n_class, n_source, n_gas, n_steps, periodic, sea_salt_opt, &
       i_na, i_cl, n_time_e, n_mode_e, n_time_b, n_mode_b, vertmix
  read(u) seed
  read(u) dt, chem_dt, dx, dy, num_particles
  n_cells = nx * nz * ny
  allocate(aero_data%density(n_spec), aero_data%name(n_spec))
  read(u) aero_data%density
  do s = 1, n_spec
     write(aero_data%name(s), '(a,i2.2)') 'S', s
  end do
  if (i_na > 0) aero_data%name(i_na) = 'Na'
  if (i_cl > 0) aero_data%name(i_cl) = 'Cl'
  allocate(aero_data%source_name(n_source), aero_data%weight_class_name(n_class), cnt(max(n_source, 1)))
  do s = 1, n_source
     write(aero_data%source_name(s), '(a,i2.2)') 'sea_salt_', s     ! (so that the sea-salt modes find theirs)
  end do
  do s = 1, n_class
     write(aero_data%weight_class_name(s), '(a,i2.2)') 'sea_salt_', s
  end do
  allocate(gas_data%name(n_gas))
  do s = 1, n_gas
     write(gas_data%name(s), '(a,i3.3)') 'G', s
  end do
  allocate(env_states(nx, nz, ny), aero_states(nx, nz, ny), scenario(nx, nz, ny), gas_states(nx, nz, ny))
  do j = 1, ny
  do k = 1, nz
  do i = 1, nx
     allocate(aero_states(i,k,j)%awa%weight(1, n_class), aero_states(i,k,j)%n_part_ideal(1, n_class))
     do s = 1, n_class
        read(u) aero_states(i,k,j)%awa%weight(1, s)%magnitude
     end do
     read(u) aero_states(i,k,j)%n_part_ideal
     aero_states(i,k,j)%apa%n_part = n_part
     allocate(aero_states(i,k,j)%apa%particle(n_part))
     do q = 1, n_part
        allocate(aero_states(i,k,j)%apa%particle(q)%vol(n_spec))
        read(u) aero_states(i,k,j)%apa%particle(q)%vol
        read(u) cnt(1:n_source)
        read(u) aero_states(i,k,j)%apa%particle(q)%weight_class
        read(u) aero_states(i,k,j)%apa%particle(q)%id
        read(u) aero_states(i,k,j)%apa%particle(q)%least_create_time, &
             aero_states(i,k,j)%apa%particle(q)%greatest_create_time
        allocate(aero_states(i,k,j)%apa%particle(q)%component(sum(cnt(1:n_source))))
        kk = 0
        do s = 1, n_source
           do m = 1, cnt(s)
              kk = kk + 1
              aero_states(i,k,j)%apa%particle(q)%component(kk)%source_id = s
              aero_states(i,k,j)%apa%particle(q)%component(kk)%create_time = &
                   aero_states(i,k,j)%apa%particle(q)%least_create_time
           end do
        end do
     end do
     allocate(gas_states(i,k,j)%mix_rat(n_gas))
     read(u) gas_states(i,k,j)%mix_rat
     read(u) has_em
     if (has_em /= 0) then
        allocate(scenario(i,k,j)%aero_emission_time(n_time_e), scenario(i,k,j)%aero_emission_rate_scale(n_time_e), &
             scenario(i,k,j)%aero_emission(n_time_e), scenario(i,k,j)%gas_emission_time(n_time_e), &
             scenario(i,k,j)%gas_emission_rate_scale(n_time_e), scenario(i,k,j)%gas_emission(n_time_e))
        read(u) scenario(i,k,j)%aero_emission_time
        read(u) scenario(i,k,j)%aero_emission_rate_scale
        scenario(i,k,j)%gas_emission_time = scenario(i,k,j)%aero_emission_time
        scenario(i,k,j)%gas_emission_rate_scale = scenario(i,k,j)%aero_emission_rate_scale
        do t = 1, n_time_e
           call read_dist(scenario(i,k,j)%aero_emission(t), n_mode_e)
           allocate(scenario(i,k,j)%gas_emission(t)%mix_rat(n_gas))
           read(u) scenario(i,k,j)%gas_emission(t)%mix_rat
        end do
     else          ! init_zero_emissions, wrf_pmc_init.F90:1436-1461
        allocate(scenario(i,k,j)%aero_emission_time(0), scenario(i,k,j)%aero_emission_rate_scale(0), &
             scenario(i,k,j)%aero_emission(0), scenario(i,k,j)%gas_emission_time(0), &
             scenario(i,k,j)%gas_emission_rate_scale(0), scenario(i,k,j)%gas_emission(0))
     end if
     allocate(scenario(i,k,j)%aero_dilution_time(n_time_b), scenario(i,k,j)%aero_dilution_rate(n_time_b), &
          scenario(i,k,j)%aero_background(n_time_b), scenario(i,k,j)%gas_dilution_time(n_time_b), &
          scenario(i,k,j)%gas_dilution_rate(n_time_b), scenario(i,k,j)%gas_background(n_time_b))
     read(u) scenario(i,k,j)%aero_dilution_time
     read(u) scenario(i,k,j)%aero_dilution_rate
     read(u) scenario(i,k,j)%gas_dilution_rate
     scenario(i,k,j)%gas_dilution_time = scenario(i,k,j)%aero_dilution_time
     do t = 1, n_time_b
        call read_dist(scenario(i,k,j)%aero_background(t), n_mode_b)
        allocate(scenario(i,k,j)%gas_background(t)%mix_rat(n_gas))
        read(u) scenario(i,k,j)%gas_background(t)%mix_rat
     end do
  end do
  end do
  end do
  close(u)

  call wrf_mock_allocate(grid, nx, nz, ny, n_class)
  grid%dt = real(dt); grid%dx = real(dx); grid%dy = real(dy); grid%rdx = 1.0 / grid%dx; grid%rdy = 1.0 / grid%dy
  grid%partmc_chem_dt = real(chem_dt); grid%num_particles = real(num_particles)
  grid%random_seed = int(seed); grid%partmc_seasalt_param = sea_salt_opt
  grid%do_coagulation = .true.; grid%do_mosaic = .false.; grid%do_transport = .true.; grid%do_deposition = .true.
  config_flags%periodic_x = periodic /= 0; config_flags%periodic_y = periodic /= 0
  config_flags%vertmix_onoff = vertmix

  allocate(n_parts(n_cells), tn(n_cells), tm(n_cells), nd(100, n_cells), md(100, n_cells))
  open(newunit=uf, file=trim(ffields), access='stream', form='unformatted', status='old')
  open(newunit=uo, file=trim(fout), access='stream', form='unformatted', status='replace')
  do step = 1, n_steps
     call read_fields()
     call partmc_timestep_pmcx(grid, config_flags, scenario, env_states, aero_data, aero_states, &
          gas_data, gas_states, 1, nx, 1, ny, 1, nz, nx, ny, nz)
     call wrf_pmc_trans_driver_pmcx(grid, config_flags, env_states, scenario, aero_data, aero_states, &
          gas_states, 1, nx, 1, ny, 1, nz, nx, ny, nz)
     write(uo) env_states%temp, env_states%pressure, env_states%rel_humid, env_states%height, &
          env_states%cell_volume, env_states%inverse_density
     write(uo) last_aer_res
     call partmc_process_pmcx(tn, tm, nd, md, n_parts=n_parts)
     write(uo) n_parts
     write(uo) last_n_samp, last_n_coag, last_n_moved, last_n_deposited
  end do
  close(uf)
  call partmc_sync_host_pmcx(aero_states, gas_states, 1, nx, 1, ny, 1, nz)
  do j = 1, ny
  do k = 1, nz
  do i = 1, nx
     do s = 1, n_class
        write(uo) aero_states(i,k,j)%awa%weight(1, s)%magnitude
     end do
     write(uo) gas_states(i,k,j)%mix_rat
     write(uo) aero_states(i,k,j)%apa%n_part
     do q = 1, aero_states(i,k,j)%apa%n_part
        write(uo) aero_states(i,k,j)%apa%particle(q)%vol
        cnt = 0
        n_comp = aero_particle_n_components(aero_states(i,k,j)%apa%particle(q))
        t_lo = huge(1d0); t_hi = -huge(1d0)
        do kk = 1, n_comp
           s = aero_states(i,k,j)%apa%particle(q)%component(kk)%source_id
           cnt(s) = cnt(s) + 1
           t_lo = min(t_lo, aero_states(i,k,j)%apa%particle(q)%component(kk)%create_time)
           t_hi = max(t_hi, aero_states(i,k,j)%apa%particle(q)%component(kk)%create_time)
        end do
        write(uo) cnt(1:n_source)
        write(uo) aero_states(i,k,j)%apa%particle(q)%weight_class
        write(uo) aero_states(i,k,j)%apa%particle(q)%id
        write(uo) aero_states(i,k,j)%apa%particle(q)%least_create_time, &
             aero_states(i,k,j)%apa%particle(q)%greatest_create_time
     end do
     ! component(:) of the cell's particles, one list after the other: int32 length, then
     ! (int32 source_id, float64 create_time) per entry
     do q = 1, aero_states(i,k,j)%apa%n_part
        n_comp = aero_particle_n_components(aero_states(i,k,j)%apa%particle(q))
        write(uo) n_comp
        do kk = 1, n_comp
           write(uo) aero_states(i,k,j)%apa%particle(q)%component(kk)%source_id
           write(uo) aero_states(i,k,j)%apa%particle(q)%component(kk)%create_time
        end do
     end do
  end do
  end do
  end do
  close(uo)
  call pmcx_shim_finalize()
  print '(a,i0,a,i0,a)', 'test_driver ok: ', n_cells, ' cells, ', n_steps, ' WRF steps'

contains

  subroutine read_dist(dist, n_mode)
    type(aero_dist_t), intent(inout) :: dist
    integer, intent(in) :: n_mode
    integer :: mm
    allocate(dist%mode(n_mode))
    do mm = 1, n_mode
       allocate(dist%mode(mm)%vol_frac(n_spec), dist%mode(mm)%vol_frac_std(n_spec))
       dist%mode(mm)%vol_frac_std = 0d0
       read(u) dist%mode(mm)%num_conc, dist%mode(mm)%char_radius, dist%mode(mm)%log10_std_dev_radius
       read(u) dist%mode(mm)%vol_frac
       read(u) dist%mode(mm)%source, dist%mode(mm)%weight_class
    end do
  end subroutine read_dist

  subroutine read_fields()
    read(uf) grid%time_since_sim_start
    read(uf) grid%p, grid%pb, grid%t_2, grid%moist, grid%alt, grid%z, grid%z_at_w, grid%xkhh, grid%exch_h
    read(uf) grid%u_2, grid%v_2, grid%u_1, grid%v_1
    read(uf) grid%u_flux, grid%v_flux, grid%w_flux
    read(uf) grid%xland, grid%u10, grid%v10, grid%msftx, grid%msfty, grid%ust, grid%rmol, grid%znt
    read(uf) grid%ivgtyp
  end subroutine read_fields

end program test_driver
