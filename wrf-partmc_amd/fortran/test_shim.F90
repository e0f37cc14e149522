! test_shim.F90 -- drives the shim the way partmc_timestep drives mc_coag
! (interface/wrf_pmc_driver.F90:169-198): a loop over cells, each an
! aero_state_t, through mc_coag_pmcx (per cell) or partmc_coag_cells_pmcx
! (batched).  Input and output are raw binary files so a test can compare the
! result bit for bit with the CPU oracle.
!
! usage: test_shim <in.bin> <out.bin> <mode: 0 per-cell, 1 batched>
! in.bin : int32 n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps;
!          int64 seed; float64 del_t; float64 density(n_spec);
!          per cell: float64 temp, pressure, mag(n_class);
!                    per particle: float64 vol(n_spec), int32 n_orig(n_source),
!                                  int32 class (1-based), int64 id, float64 lct, gct
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
  integer :: c, i, s, step, u

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
  allocate(env_states(n_cells), aero_states(n_cells), n_samp(n_cells), n_coag(n_cells))
  do c = 1, n_cells
     read(u) env_states(c)%temp, env_states(c)%pressure
     allocate(aero_states(c)%awa%weight(1, n_class))
     do s = 1, n_class
        read(u) aero_states(c)%awa%weight(1, s)%magnitude
     end do
     aero_states(c)%apa%n_part = n_part
     allocate(aero_states(c)%apa%particle(n_part))
     do i = 1, n_part
        allocate(aero_states(c)%apa%particle(i)%vol(n_spec))
        allocate(aero_states(c)%apa%particle(i)%n_orig_part(n_source))
        read(u) aero_states(c)%apa%particle(i)%vol
        read(u) aero_states(c)%apa%particle(i)%n_orig_part
        read(u) aero_states(c)%apa%particle(i)%weight_class
        read(u) aero_states(c)%apa%particle(i)%id
        read(u) aero_states(c)%apa%particle(i)%least_create_time, &
             aero_states(c)%apa%particle(i)%greatest_create_time
     end do
  end do
  close(u)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  do step = 0, n_steps - 1
     if (mode == 1) then
        call pmcx_set_rng(seed, step, 0)
        call partmc_coag_cells_pmcx(kernel, env_states, aero_data, aero_states, del_t, &
             n_samp, n_coag)
     else
        ! the reference's loop shape: one mc_coag call per cell (driver:169-198)
        do c = 1, n_cells
           call pmcx_set_rng(seed, step, c - 1)
           call pmcx_shim_finalize()   ! cell id changes: new context keyed by it
           call mc_coag_pmcx(kernel, env_states(c), aero_data, aero_states(c), del_t, &
                n_samp(c), n_coag(c))
        end do
     end if
     write(u) n_samp, n_coag
  end do
  do c = 1, n_cells
     write(u) aero_states(c)%apa%n_part
     do i = 1, aero_states(c)%apa%n_part
        write(u) aero_states(c)%apa%particle(i)%vol
        write(u) aero_states(c)%apa%particle(i)%n_orig_part
        write(u) aero_states(c)%apa%particle(i)%weight_class
        write(u) aero_states(c)%apa%particle(i)%id
        write(u) aero_states(c)%apa%particle(i)%least_create_time, &
             aero_states(c)%apa%particle(i)%greatest_create_time
     end do
  end do
  close(u)
  call pmcx_shim_finalize()
  print '(a,i0,a,i0)', 'test_shim ok: cells=', n_cells, ' events=', sum(n_coag)
end program test_shim
