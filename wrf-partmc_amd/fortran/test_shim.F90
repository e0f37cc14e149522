! test_shim.F90 -- drives the shim the way partmc_timestep drives mc_coag
! (interface/wrf_pmc_driver.F90:169-198): a loop over cells, each an
! aero_state_t, through mc_coag_pmcx (per cell) or partmc_coag_cells_pmcx
! (batched).  Input and output are raw binary files so a test can compare the
! result bit for bit with the CPU oracle.
!
! usage: test_shim <in.bin> <out.bin> <mode: 0 per-cell, 1 batched, 2 resident, 3 resident from fields> [fields.bin]
! mode 3: as mode 2, but the transport is driven from WRF-shaped fields (fields.bin: float64 dt, dx,
!         dy; float32 alt, xkhh (nx+2, nz, ny+2), u_flux (nx+1, nz, ny, n_class), v_flux (nx, nz,
!         ny+1, n_class), w_flux (nx, nz+1, ny, n_class), z (nx, nz, ny), z_at_w, exch_h (nx, nz+1,
!         ny)) through trans_from_fields_cells_pmcx, followed by seasalt_cells_pmcx on level 0
!         (the per-step record is then n_moved, n_arrived, n_emit, n_samp, n_coag)
! mode 2: the cells are a 2 x 2 x (n_cells / 4) grid (x, level, y), uploaded once; every step
!         runs transport, dry deposition on level 0 and mc_coag on the device; one download at
!         the end (the per-step record is then n_moved, n_arrived, n_removed, n_samp, n_coag)
! in.bin : int32 n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps;
!          int64 seed; float64 del_t; float64 density(n_spec);
!          per cell: float64 temp, pressure, mag(n_class), n_part_ideal(n_class);
!                    per particle: float64 vol(n_spec), int32 n_orig(n_source),
!                                  int32 class (1-based), int64 id, float64 lct, gct
! Every particle is given component(:) from its source counts (create_time = lct) and marks in
! the fields the device does not carry (water_hyst_leg = mod(id, 7) + 1, core_vol and
! absorb_cross_sect(:) = id): the output shows where they ended up.
! out.bin: per step the counters; then per cell n_part and per particle the input record followed
!          by int32 water_hyst_leg, float64 core_vol, float64 absorb_cross_sect(1), int32 n_components,
!          float64 min and max of component(:)%create_time; behind a cell's particles their
!          component(:) lists one after the other, (int32 source_id, float64 create_time) per entry
program test_shim
  use iso_c_binding
  use pmc_mock
  use pmcx_shim
  implicit none
  character(len=512) :: fin, fout, smode
  integer :: n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps, mode
  integer(kind=8) :: seed
  real(kind=dp) :: del_t
  type(aero_data_t) :: aero_data
  type(env_state_t), allocatable :: env_states(:)
  type(aero_state_t), allocatable :: aero_states(:)
  integer, allocatable :: n_samp(:), n_coag(:)
  integer :: c, i, s, step, u, q, k, nx, nz, ny, lev, n_comp
  integer, allocatable :: cnt(:)
  real(kind=dp) :: t_lo, t_hi
  integer, allocatable :: n_moved(:), n_arrived(:), n_removed(:), surface(:), n_thinned(:)
  real(kind=dp), allocatable :: expected(:,:)
  character(len=512) :: ffields
  integer :: uf
  real(kind=dp) :: f_dt, f_dx, f_dy
  real(c_float), allocatable :: f_alt(:,:,:), f_xkhh(:,:,:), f_u(:,:,:,:), f_v(:,:,:,:), f_w(:,:,:,:), &
       f_z(:,:,:), f_zw(:,:,:), f_exch(:,:,:)
  integer, allocatable :: n_emit(:), ocean(:)
  real(kind=dp), allocatable :: wind(:)
  real(kind=dp), allocatable :: prob_h(:,:,:), prob_v(:,:,:), cell_volume(:), ra(:), ustar(:), &
       alpha(:), big_a(:), height(:), inv_density(:)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, smode)
  read(smode, *) mode

  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps
  read(u) seed
  read(u) del_t
  allocate(aero_data%density(n_spec))
  read(u) aero_data%density
  allocate(aero_data%source_name(n_source), cnt(max(n_source, 1)))
  do s = 1, n_source
     write(aero_data%source_name(s), '(a,i0)') 'source_', s
  end do
  allocate(env_states(n_cells), aero_states(n_cells), n_samp(n_cells), n_coag(n_cells))
  do c = 1, n_cells
     read(u) env_states(c)%temp, env_states(c)%pressure
     allocate(aero_states(c)%awa%weight(1, n_class))
     do s = 1, n_class
        read(u) aero_states(c)%awa%weight(1, s)%magnitude
     end do
     allocate(aero_states(c)%n_part_ideal(1, n_class))
     read(u) aero_states(c)%n_part_ideal
     aero_states(c)%apa%n_part = n_part
     allocate(aero_states(c)%apa%particle(n_part))
     do i = 1, n_part
        allocate(aero_states(c)%apa%particle(i)%vol(n_spec))
        read(u) aero_states(c)%apa%particle(i)%vol
        read(u) cnt(1:n_source)
        read(u) aero_states(c)%apa%particle(i)%weight_class
        read(u) aero_states(c)%apa%particle(i)%id
        read(u) aero_states(c)%apa%particle(i)%least_create_time, &
             aero_states(c)%apa%particle(i)%greatest_create_time
        allocate(aero_states(c)%apa%particle(i)%component(sum(cnt(1:n_source))))
        k = 0
        do s = 1, n_source
           do q = 1, cnt(s)
              k = k + 1
              aero_states(c)%apa%particle(i)%component(k)%source_id = s
              aero_states(c)%apa%particle(i)%component(k)%create_time = &
                   aero_states(c)%apa%particle(i)%least_create_time
           end do
        end do
        aero_states(c)%apa%particle(i)%water_hyst_leg = &
             int(mod(aero_states(c)%apa%particle(i)%id, 7_8)) + 1
        aero_states(c)%apa%particle(i)%core_vol = real(aero_states(c)%apa%particle(i)%id, dp)
        allocate(aero_states(c)%apa%particle(i)%absorb_cross_sect(n_swbands))
        aero_states(c)%apa%particle(i)%absorb_cross_sect = real(aero_states(c)%apa%particle(i)%id, dp)
     end do
  end do
  close(u)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  if (mode == 3) then
     nx = 2; nz = 2; ny = n_cells / 4
     call get_command_argument(4, ffields)
     open(newunit=uf, file=trim(ffields), access='stream', form='unformatted', status='old')
     allocate(f_alt(nx + 2, nz, ny + 2), f_xkhh(nx + 2, nz, ny + 2), f_u(nx + 1, nz, ny, n_class), &
          f_v(nx, nz, ny + 1, n_class), f_w(nx, nz + 1, ny, n_class), f_z(nx, nz, ny), &
          f_zw(nx, nz + 1, ny), f_exch(nx, nz + 1, ny))
     read(uf) f_dt, f_dx, f_dy
     read(uf) f_alt, f_xkhh, f_u, f_v, f_w, f_z, f_zw, f_exch
     close(uf)
     allocate(n_moved(n_cells), n_arrived(n_cells), n_emit(n_cells), ocean(n_cells), wind(n_cells), &
          cell_volume(n_cells), height(n_cells))
     do c = 1, n_cells
        lev = mod((c - 1) / nx, nz)
        ocean(c) = merge(1, 0, lev == 0)
        wind(c) = 6d0 + 0.5d0 * c
        height(c) = 40d0
        cell_volume(c) = 1d0
     end do
     call pmcx_set_rng(seed, 0, 0)
     call pmcx_upload_states(env_states, aero_data, aero_states, 2 * n_part)
     do step = 0, n_steps - 1
        call pmcx_set_rng(seed, step, 0)
        call trans_from_fields_cells_pmcx(nx, nz, ny, .true., .true., f_dt, f_dx, f_dy, 1, f_alt, f_xkhh, &
             f_u, f_v, f_w, f_z, f_zw, f_exch, cell_volume, n_moved, n_arrived)
        call seasalt_cells_pmcx(del_t, del_t * step, 1, 1, 2, ocean, wind, height, (/1, 2/), (/1, 2/), n_emit)
        call mc_coag_resident_pmcx(kernel, del_t, n_samp, n_coag)
        call aero_state_rebalance_cells_pmcx(.true., .true.)
        write(u) n_moved, n_arrived, n_emit, n_samp, n_coag
     end do
     call pmcx_download_states(aero_states)
  end if
  if (mode == 2) then
     nx = 2; nz = 2; ny = n_cells / 4
     allocate(n_moved(n_cells), n_arrived(n_cells), n_removed(n_cells), surface(n_cells))
     allocate(expected(n_class, n_cells), n_thinned(n_cells))
     allocate(prob_h(4, n_class, n_cells), prob_v(nz, n_class, n_cells), cell_volume(n_cells))
     allocate(ra(n_cells), ustar(n_cells), alpha(n_cells), big_a(n_cells), height(n_cells), &
          inv_density(n_cells))
     do c = 1, n_cells
        lev = mod((c - 1) / nx, nz)               ! c - 1 = (j * nz + k) * nx + i
        surface(c) = merge(1, 0, lev == 0)
        do s = 1, n_class
           do q = 1, 4
              prob_h(q, s, c) = 0.02d0 + 0.005d0 * (q - 1) + 0.001d0 * (s - 1)
           end do
           do k = 1, nz
              prob_v(k, s, c) = merge(0d0, 0.01d0, k - 1 == lev)
           end do
        end do
        cell_volume(c) = 1d0
        ra(c) = 30d0 + c
        ustar(c) = 0.3d0
        alpha(c) = 1.2d0
        big_a(c) = 2d-3
        height(c) = 40d0
        inv_density(c) = 0.8d0
     end do
     call pmcx_set_rng(seed, 0, 0)
     call pmcx_upload_states(env_states, aero_data, aero_states, 2 * n_part)
     do step = 0, n_steps - 1
        call pmcx_set_rng(seed, step, 0)
        call trans_preweight_cells_pmcx(nx, nz, ny, .true., .true., prob_h, prob_v, cell_volume, &
             expected, n_thinned)
        call trans_cells_pmcx(nx, nz, ny, .true., .true., prob_h, prob_v, cell_volume, &
             n_moved, n_arrived)
        call dry_dep_cells_pmcx(del_t, surface, ra, ustar, alpha, big_a, height, inv_density, &
             n_removed)
        call mc_coag_resident_pmcx(kernel, del_t, n_samp, n_coag)
        call aero_state_rebalance_cells_pmcx(.true., .true.)
        write(u) n_thinned, n_moved, n_arrived, n_removed, n_samp, n_coag
        write(u) expected
     end do
     call pmcx_download_states(aero_states)
  end if
  do step = 0, merge(-1, n_steps - 1, mode >= 2)
     if (mode == 1) then
        call pmcx_set_rng(seed, step, 0)
        call partmc_coag_cells_pmcx(kernel, env_states, aero_data, aero_states, del_t, &
             n_samp, n_coag)
     else
        ! the reference's loop shape: one mc_coag call per cell (driver:169-198)
        do c = 1, n_cells
           call pmcx_set_rng(seed, step, c - 1)   ! the documented per-cell sequence (INTEGRATION.md 2a)
           call mc_coag_pmcx(kernel, env_states(c), aero_data, aero_states(c), del_t, &
                n_samp(c), n_coag(c))
        end do
     end if
     ! aero_state_rebalance on the host states (driver:248-250)
     call pmcx_set_rng(seed, step, 0)
     call aero_state_rebalance_pmcx(env_states, aero_data, aero_states, .true., .true.)
     write(u) n_samp, n_coag
  end do
  do c = 1, n_cells
     write(u) aero_states(c)%apa%n_part
     do i = 1, aero_states(c)%apa%n_part
        write(u) aero_states(c)%apa%particle(i)%vol
        cnt = 0
        n_comp = aero_particle_n_components(aero_states(c)%apa%particle(i))
        t_lo = huge(1d0); t_hi = -huge(1d0)
        do k = 1, n_comp
           s = aero_states(c)%apa%particle(i)%component(k)%source_id
           cnt(s) = cnt(s) + 1
           t_lo = min(t_lo, aero_states(c)%apa%particle(i)%component(k)%create_time)
           t_hi = max(t_hi, aero_states(c)%apa%particle(i)%component(k)%create_time)
        end do
        write(u) cnt(1:n_source)
        write(u) aero_states(c)%apa%particle(i)%weight_class
        write(u) aero_states(c)%apa%particle(i)%id
        write(u) aero_states(c)%apa%particle(i)%least_create_time, &
             aero_states(c)%apa%particle(i)%greatest_create_time
        write(u) aero_states(c)%apa%particle(i)%water_hyst_leg
        write(u) aero_states(c)%apa%particle(i)%core_vol
        write(u) aero_states(c)%apa%particle(i)%absorb_cross_sect(1)
        write(u) n_comp
        write(u) t_lo, t_hi
     end do
     ! ... and the lists themselves, particle after particle
     do i = 1, aero_states(c)%apa%n_part
        do k = 1, aero_particle_n_components(aero_states(c)%apa%particle(i))
           write(u) aero_states(c)%apa%particle(i)%component(k)%source_id
           write(u) aero_states(c)%apa%particle(i)%component(k)%create_time
        end do
     end do
  end do
  close(u)
  call pmcx_shim_finalize()
  print '(a,i0,a,i0)', 'test_shim ok: cells=', n_cells, ' events=', sum(n_coag)
end program test_shim
