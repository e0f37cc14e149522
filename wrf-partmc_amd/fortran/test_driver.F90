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
! usage: test_driver <state.bin> <fields.bin> <out.bin> [wrap [px py a b exchange_dir]]
!   wrap = 1: the halo cells of grid%'s fields are filled by periodic wrap after reading (what WRF's
!             periodic halo exchange leaves there); 0: by the nearest cell of the domain
!   px py a b exchange_dir: this process is rank (a, b) of a px x py decomposition of the domain (WRF's
!             patches: columns a nx/px + 1 .. (a + 1) nx/px, rows likewise); it reads the whole case,
!             keeps its patch, hands the two drivers the patch's bounds exactly as WRF would, and reaches
!             its neighbours through files in exchange_dir (the stand-in for MPI_Sendrecv:
!             pmcx_shim_set_neighbours).  out.bin then holds this rank's cells only.
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
module rank_exchange
  ! The file-backed stand-in for MPI_Sendrecv between the ranks of a cut domain: a message is a file
  ! <dir>/m<sequence>_<direction>_<from>_<to>.bin, written whole and then announced by an empty .ok
  ! file; the receiver polls for the announcement, reads and deletes both.
  use iso_c_binding
  implicit none
  character(len=512), save :: xdir = ''
  integer, save :: px = 1, py = 1, ra = 0, rb = 0, seq = 0
  logical, save :: periodic_dom = .false.
contains
  integer function neighbour(side)          ! rank on side 0 south, 1 north, 2 west, 3 east; -1: none
    integer, intent(in) :: side
    integer :: a, b
    a = ra; b = rb
    select case (side)
    case (0); b = b - 1
    case (1); b = b + 1
    case (2); a = a - 1
    case (3); a = a + 1
    end select
    neighbour = -1
    if (periodic_dom) then
       a = modulo(a, px); b = modulo(b, py)
    else if (a < 0 .or. a >= px .or. b < 0 .or. b >= py) then
       return
    end if
    if (a == ra .and. b == rb) return      ! (one rank across that direction: no cut)
    neighbour = b * px + a
  end function neighbour

  subroutine file_shift(dir, send, n_send, recv, n_recv)
    integer, intent(in) :: dir
    type(c_ptr), value :: send, recv
    integer(c_int64_t), value :: n_send, n_recv
    integer(c_int8_t), pointer :: bs(:), br(:)
    character(len=700) :: name
    integer :: u, me, to, from, tries
    logical :: there
    seq = seq + 1
    me = rb * px + ra
    to = neighbour(dir); from = neighbour(ieor(dir, 1))
    if (n_send > 0 .and. to >= 0) then
       call c_f_pointer(send, bs, [n_send])
       write(name, '(a,a,i8.8,a,i1,a,i3.3,a,i3.3)') trim(xdir), '/m', seq, '_', dir, '_', me, '_', to
       open(newunit=u, file=trim(name) // '.bin', access='stream', form='unformatted', status='replace')
       write(u) bs
       close(u)
       open(newunit=u, file=trim(name) // '.ok', status='replace')
       close(u)
    end if
    if (n_recv > 0 .and. from >= 0) then
       call c_f_pointer(recv, br, [n_recv])
       write(name, '(a,a,i8.8,a,i1,a,i3.3,a,i3.3)') trim(xdir), '/m', seq, '_', dir, '_', from, '_', me
       tries = 0
       do
          inquire(file=trim(name) // '.ok', exist=there)
          if (there) exit
          tries = tries + 1
          if (tries > 600000) then
             write(0, '(a,a)') 'rank_exchange: no message ', trim(name)
             stop 3
          end if
          call execute_command_line('sleep 0.002')
       end do
       open(newunit=u, file=trim(name) // '.bin', access='stream', form='unformatted', status='old')
       read(u) br
       close(u, status='delete')
       open(newunit=u, file=trim(name) // '.ok', status='old')
       close(u, status='delete')
    end if
  end subroutine file_shift
end module rank_exchange

program test_driver
  use iso_c_binding
  use pmc_mock
  use wrf_mock
  use pmcx_shim
  use rank_exchange
  implicit none
  character(len=512) :: fstate, ffields, fout, arg
  integer :: wrap, is, ie, js, je, hal
  logical :: ranked
  type(env_state_t), allocatable :: env_p(:,:,:)
  type(aero_state_t), allocatable :: aero_p(:,:,:)
  type(scenario_t), allocatable :: scn_p(:,:,:)
  type(gas_state_t), allocatable :: gas_p(:,:,:)
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
  ! (environment, for scripts/fuzz_driver.py: PMCX_TEST_HEADROOM = slots per cell / ideal count the block
  ! starts with; PMCX_TEST_AUTO_GROW = 0: a full cell stops the run as until round 4)
  call get_environment_variable('PMCX_TEST_HEADROOM', arg, status=i)
  if (i == 0 .and. len_trim(arg) > 0) then
     read(arg, *) t_lo
     call pmcx_shim_set_headroom(t_lo)
  end if
  call get_environment_variable('PMCX_TEST_AUTO_GROW', arg, status=i)
  if (i == 0 .and. len_trim(arg) > 0) then
     read(arg, *) j
     call pmcx_shim_set_auto_grow(j /= 0)
  end if
  wrap = 0; ranked = .false.
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg); read(arg, *) wrap
  end if
  if (command_argument_count() >= 9) then
     ranked = .true.
     call get_command_argument(5, arg); read(arg, *) px
     call get_command_argument(6, arg); read(arg, *) py
     call get_command_argument(7, arg); read(arg, *) ra
     call get_command_argument(8, arg); read(arg, *) rb
     call get_command_argument(9, xdir)
  end if
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

  ! this process's patch (the whole domain unless it is a rank of a cut one)
  is = 1; ie = nx; js = 1; je = ny; hal = 1
  if (ranked) then
     if (mod(nx, px) /= 0 .or. mod(ny, py) /= 0) stop 'test_driver: px, py must divide nx, ny'
     is = ra * (nx / px) + 1; ie = (ra + 1) * (nx / px); js = rb * (ny / py) + 1; je = (rb + 1) * (ny / py)
     hal = 3
     periodic_dom = periodic /= 0
     call pmcx_shim_set_neighbours(file_shift, neighbour(0) >= 0, neighbour(1) >= 0, neighbour(2) >= 0, &
          neighbour(3) >= 0)
  end if
  allocate(env_p(is:ie, nz, js:je), aero_p(is:ie, nz, js:je), scn_p(is:ie, nz, js:je), gas_p(is:ie, nz, js:je))
  do j = js, je
  do k = 1, nz
  do i = is, ie
     env_p(i,k,j) = env_states(i,k,j); aero_p(i,k,j) = aero_states(i,k,j)
     scn_p(i,k,j) = scenario(i,k,j); gas_p(i,k,j) = gas_states(i,k,j)
  end do
  end do
  end do
  n_cells = (ie - is + 1) * nz * (je - js + 1)
  call wrf_mock_allocate(grid, nx, nz, ny, n_class, hal)
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
     call partmc_timestep_pmcx(grid, config_flags, scn_p, env_p, aero_data, aero_p, &
          gas_data, gas_p, is, ie, js, je, 1, nz, nx, ny, nz)
     call wrf_pmc_trans_driver_pmcx(grid, config_flags, env_p, scn_p, aero_data, aero_p, &
          gas_p, is, ie, js, je, 1, nz, nx, ny, nz)
     write(uo) env_p%temp, env_p%pressure, env_p%rel_humid, env_p%height, &
          env_p%cell_volume, env_p%inverse_density
     if (ranked) then          ! (the block's resistances: this rank's cells of them)
        write(uo) (((last_aer_res(((j - js + 1) * nz + (k - 1)) * (ie - is + 3) + (i - is + 1) + 1), i = is, ie), k = 1, nz), j = js, je)
     else
        write(uo) last_aer_res
     end if
     call partmc_process_pmcx(tn, tm, nd, md, n_parts=n_parts)
     write(uo) n_parts
     write(uo) last_n_samp, last_n_coag, last_n_moved, last_n_deposited
  end do
  close(uf)
  call partmc_sync_host_pmcx(aero_p, gas_p, is, ie, js, je, 1, nz)
  do j = js, je
  do k = 1, nz
  do i = is, ie
     aero_states(i,k,j) = aero_p(i,k,j); gas_states(i,k,j) = gas_p(i,k,j)
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
  call pmcx_shim_growth(i, j)
  call pmcx_shim_finalize()
  print '(a,i0,a,i0,a,i0,a,i0,a)', 'test_driver ok: ', n_cells, ' cells, ', n_steps, ' WRF steps; ', i, &
       ' slots a cell, the block grew ', j, ' times'

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

  ! the file holds every array with one halo cell (0 : n + 1; the staggered ones 0 : n + 2); a rank of a
  ! cut domain keeps more halo, filled like the first: by periodic wrap or from the nearest cell
  subroutine read_fields()
    real, allocatable :: a3(:,:,:), b3(:,:,:), f4(:,:,:,:), a2(:,:)
    integer, allocatable :: i2(:,:)
    integer :: q
    read(uf) grid%time_since_sim_start
    allocate(a3(0:nx+1, nz+1, 0:ny+1), b3(0:nx+2, nz+1, 0:ny+2), f4(0:nx+2, nz+1, 0:ny+2, n_class + 1), &
         a2(0:nx+1, 0:ny+1), i2(0:nx+1, 0:ny+1))
    read(uf) a3; call put3(grid%p, a3, 0)
    read(uf) a3; call put3(grid%pb, a3, 0)
    read(uf) a3; call put3(grid%t_2, a3, 0)
    read(uf) a3; call put3(grid%moist(:,:,:,1), a3, 0)
    read(uf) a3; call put3(grid%alt, a3, 0)
    read(uf) a3; call put3(grid%z, a3, 0)
    read(uf) a3; call put3(grid%z_at_w, a3, 0)
    read(uf) a3; call put3(grid%xkhh, a3, 0)
    read(uf) a3; call put3(grid%exch_h, a3, 0)
    read(uf) b3; call put3(grid%u_2, b3, 1)
    read(uf) b3; call put3(grid%v_2, b3, 1)
    read(uf) b3; call put3(grid%u_1, b3, 1)
    read(uf) b3; call put3(grid%v_1, b3, 1)
    read(uf) f4
    do q = 1, n_class + 1
       call put3(grid%u_flux(:,:,:,q), f4(:,:,:,q), 1)
    end do
    read(uf) f4
    do q = 1, n_class + 1
       call put3(grid%v_flux(:,:,:,q), f4(:,:,:,q), 1)
    end do
    read(uf) f4
    do q = 1, n_class + 1
       call put3(grid%w_flux(:,:,:,q), f4(:,:,:,q), 1)
    end do
    read(uf) a2; call put2(grid%xland, a2)
    read(uf) a2; call put2(grid%u10, a2)
    read(uf) a2; call put2(grid%v10, a2)
    read(uf) a2; call put2(grid%msftx, a2)
    read(uf) a2; call put2(grid%msfty, a2)
    read(uf) a2; call put2(grid%ust, a2)
    read(uf) a2; call put2(grid%rmol, a2)
    read(uf) a2; call put2(grid%znt, a2)
    read(uf) i2
    call puti(grid%ivgtyp, i2)
  end subroutine read_fields

  ! source index of a halo cell: itself inside the file's range unless wrap asks for the periodic image
  integer function src_index(g, n, stag)
    integer, intent(in) :: g, n, stag
    if (wrap /= 0) then
       if (g >= 1 .and. g <= n + stag) then
          src_index = g
       else
          src_index = modulo(g - 1, n) + 1
       end if
    else
       src_index = min(max(g, 0), n + 1 + stag)
    end if
  end function src_index

  subroutine put3(dst, src, stag)
    real, intent(inout) :: dst(1-hal:, :, 1-hal:)
    real, intent(in) :: src(0:, :, 0:)
    integer, intent(in) :: stag
    integer :: gi, gj
    do gj = lbound(dst, 3), ubound(dst, 3)
       do gi = lbound(dst, 1), ubound(dst, 1)
          dst(gi, :, gj) = src(src_index(gi, nx, stag), :, src_index(gj, ny, stag))
       end do
    end do
  end subroutine put3

  subroutine put2(dst, src)
    real, intent(inout) :: dst(1-hal:, 1-hal:)
    real, intent(in) :: src(0:, 0:)
    integer :: gi, gj
    do gj = lbound(dst, 2), ubound(dst, 2)
       do gi = lbound(dst, 1), ubound(dst, 1)
          dst(gi, gj) = src(src_index(gi, nx, 0), src_index(gj, ny, 0))
       end do
    end do
  end subroutine put2

  subroutine puti(dst, src)
    integer, intent(inout) :: dst(1-hal:, 1-hal:)
    integer, intent(in) :: src(0:, 0:)
    integer :: gi, gj
    do gj = lbound(dst, 2), ubound(dst, 2)
       do gi = lbound(dst, 1), ubound(dst, 1)
          dst(gi, gj) = src(src_index(gi, nx, 0), src_index(gj, ny, 0))
       end do
    end do
  end subroutine puti

end program test_driver
