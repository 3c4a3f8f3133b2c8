!> icar_hip_step_demo.f90 -- the whole sub-step loop of time_step.f90:440-551 driven from Fortran through the C ABI,
!! one image, configurations 1-4 (upwind/MPDATA + mp_simple): per step
!!   compute_dt (CFL reduction on the device) -> diagnostic_update -> mp() -> advect -> apply_forcing,
!! with enforce_limits in the last two sub-steps (:537-539).  Reads the tile written by
!! tests/test_gpu_fortran_host.py (raw REAL(4), Fortran order) and writes the prognostic fields back.
program icar_hip_step_demo
  use iso_c_binding
  use icar_hip
  implicit none
  type(hip_ctx_t) :: ctx
  integer :: nx, nz, ny, u, i, nsteps
  real :: dx, cfl
  double precision :: end_time, now, dt, last_mp, mp_dt
  logical :: first_mp = .true.
  real(c_float), allocatable, target :: a(:,:,:), au(:,:,:), av(:,:,:), dz_levels(:)
  real(c_double), allocatable, target :: acc(:,:)
  character(len=512) :: dir
  integer, parameter :: n3 = 15
  integer(c_int), parameter :: f3(n3) = [ICAR_F_W, ICAR_F_PRESSURE, ICAR_F_EXNER, ICAR_F_DENSITY, ICAR_F_DZ_MASS, ICAR_F_JACOBIAN, &
       ICAR_F_JACOBIAN_W, ICAR_F_ADVECTION_DZ, ICAR_F_DZDX, ICAR_F_DZDY, ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, &
       ICAR_F_POTENTIAL_TEMPERATURE]
  character(len=24), parameter :: names(n3) = [character(len=24) :: "w", "pressure", "exner", "density", "dz_mass", "jacobian", &
       "jacobian_w", "advection_dz", "dzdx", "dzdy", "water_vapor", "cloud_water", "rain", "snow", "potential_temperature"]
  integer(c_int), parameter :: adv(5) = [ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, ICAR_F_POTENTIAL_TEMPERATURE]
  ! forced variables: qv and theta relax on the boundary ring, u, v, pressure and w over the whole field
  integer(c_int), parameter :: forced(6) = [ICAR_F_WATER_VAPOR, ICAR_F_POTENTIAL_TEMPERATURE, ICAR_F_U, ICAR_F_V, ICAR_F_PRESSURE, ICAR_F_W]
  logical, parameter :: fb(6) = [.true., .true., .false., .false., .false., .false.]

  call get_command_argument(1, dir)
  open(newunit=u, file=trim(dir)//"/meta.txt", status="old"); read(u,*) nx, nz, ny, end_time, dx; close(u)
  allocate(a(nx,nz,ny), au(nx+1,nz,ny), av(nx,nz,ny+1), acc(nx,ny), dz_levels(nz))
  open(newunit=u, file=trim(dir)//"/dz_levels.bin", access="stream", form="unformatted", status="old"); read(u) dz_levels; close(u)
  call hip_create(ctx, 0, 1, nx, 1, nz, 1, ny)
  do i = 1, n3
     if (f3(i) == ICAR_F_DZDX .or. f3(i) == ICAR_F_DZDY) cycle
     call rd(trim(dir)//"/"//trim(names(i))//".bin", a); call hip_upload(ctx, f3(i), a)
  end do
  call rd(trim(dir)//"/u.bin", au);          call hip_upload(ctx, ICAR_F_U, au)
  call rd(trim(dir)//"/jacobian_u.bin", au); call hip_upload(ctx, ICAR_F_JACOBIAN_U, au)
  call rd(trim(dir)//"/dzdx.bin", au);       call hip_upload(ctx, ICAR_F_DZDX, au)
  call rd(trim(dir)//"/dqdt_u.bin", au);     call hip_dqdt_upload(ctx, ICAR_F_U, au)
  call rd(trim(dir)//"/v.bin", av);          call hip_upload(ctx, ICAR_F_V, av)
  call rd(trim(dir)//"/jacobian_v.bin", av); call hip_upload(ctx, ICAR_F_JACOBIAN_V, av)
  call rd(trim(dir)//"/dzdy.bin", av);       call hip_upload(ctx, ICAR_F_DZDY, av)
  call rd(trim(dir)//"/dqdt_v.bin", av);     call hip_dqdt_upload(ctx, ICAR_F_V, av)
  call rd(trim(dir)//"/dqdt_water_vapor.bin", a);           call hip_dqdt_upload(ctx, ICAR_F_WATER_VAPOR, a)
  call rd(trim(dir)//"/dqdt_potential_temperature.bin", a); call hip_dqdt_upload(ctx, ICAR_F_POTENTIAL_TEMPERATURE, a)
  call rd(trim(dir)//"/dqdt_pressure.bin", a);              call hip_dqdt_upload(ctx, ICAR_F_PRESSURE, a)
  call rd(trim(dir)//"/dqdt_w.bin", a);                     call hip_dqdt_upload(ctx, ICAR_F_W, a)

  now = 0.0d0; nsteps = 0
  do while (now < end_time)                                              ! time_step.f90:462
     cfl = hip_max_courant(ctx, dx, dz_levels)                           ! compute_dt :264-289, strictness 3
     dt = dble(0.9 / cfl)                                                ! cfl_reduction_factor / max, REAL(4) then widened
     dt = min(dt, 120.0d0)                                               ! update_dt :417
     if (now + dt > end_time) dt = end_time - now                        ! :469-471
     call hip_diagnostic_update(ctx)                                     ! :474
     if (dt > 1d-3) then                                                 ! :483
        if (first_mp) last_mp = now - dt                                 ! mp_driver.f90:698-713: mp_dt is the time since the last call
        first_mp = .false.
        mp_dt = now - last_mp; last_mp = now
        call hip_mp_simple(ctx, real(mp_dt), 2, nx-1, 2, ny-1, 1, nz)    ! mp() :512-523 (one image: whole tile)
        call hip_advect(ctx, 2, 2, .true., .false., real(dt), dx, adv)   ! :529
        call hip_apply_forcing(ctx, dt, forced, fb, .true., .true., .true., .true.)   ! :534
        if ((end_time - now) < dt*2) call hip_enforce_limits(ctx, adv)   ! :537-539
     end if
     now = now + dt; nsteps = nsteps + 1
  end do
  do i = 11, n3
     call hip_download(ctx, f3(i), a); call wr(trim(dir)//"/out_"//trim(names(i))//".bin", a)
  end do
  call hip_download(ctx, ICAR_F_W_REAL, a); call wr(trim(dir)//"/out_w_real.bin", a)
  call hip_download(ctx, ICAR_F_U, au); call wr(trim(dir)//"/out_u.bin", au)
  call hip_download_2dd(ctx, ICAR_F_PRECIPITATION, acc)
  open(newunit=u, file=trim(dir)//"/out_precip.bin", access="stream", form="unformatted", status="replace"); write(u) acc; close(u)
  call hip_destroy(ctx)
  print *, "icar_hip_step_demo: ok", nsteps
contains
  subroutine rd(fn, x)
    character(len=*), intent(in) :: fn
    real(c_float), intent(out) :: x(:,:,:)
    integer :: uu
    open(newunit=uu, file=fn, access="stream", form="unformatted", status="old"); read(uu) x; close(uu)
  end subroutine
  subroutine wr(fn, x)
    character(len=*), intent(in) :: fn
    real(c_float), intent(in) :: x(:,:,:)
    integer :: uu
    open(newunit=uu, file=fn, access="stream", form="unformatted", status="replace"); write(uu) x; close(uu)
  end subroutine
end program
