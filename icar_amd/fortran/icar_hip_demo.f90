!> icar_hip_demo.f90 -- Fortran host driving the device hot path through the C ABI.
!! Reads a tile written by tests/test_gpu_fortran_host.py (raw little-endian REAL(4), Fortran order),
!! runs `nsteps` x [mp_simple on the interior tile -> advection (scheme from meta.txt: 1 upwind, 2 MPDATA, 3 MPDATA in the reference's operation order = hip_mpdata_exact) of the 5 mp_simple scalars]
!! exactly as time_step.f90:512-529 orders them for one image, and writes the fields back.
program icar_hip_demo
  use iso_c_binding
  use icar_hip
  implicit none
  type(hip_ctx_t) :: ctx
  integer :: nx, nz, ny, nsteps, u, s, i, scheme
  real :: dt, dx
  real(c_float), allocatable, target :: a(:,:,:), au(:,:,:), av(:,:,:)
  real(c_double), allocatable, target :: acc(:,:)
  character(len=512) :: dir
  integer(c_int), parameter :: f3(15) = [ICAR_F_W, ICAR_F_PRESSURE, ICAR_F_EXNER, ICAR_F_DENSITY, ICAR_F_DZ_MASS, ICAR_F_JACOBIAN, &
       ICAR_F_JACOBIAN_W, ICAR_F_ADVECTION_DZ, ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, &
       ICAR_F_POTENTIAL_TEMPERATURE, ICAR_F_CLOUD_ICE, ICAR_F_GRAUPEL]
  character(len=24), parameter :: n3(15) = [character(len=24) :: "w", "pressure", "exner", "density", "dz_mass", "jacobian", &
       "jacobian_w", "advection_dz", "water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel"]
  integer(c_int), parameter :: adv(5) = [ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, ICAR_F_POTENTIAL_TEMPERATURE]

  call get_command_argument(1, dir)
  open(newunit=u, file=trim(dir)//"/meta.txt", status="old"); read(u,*) nx, nz, ny, nsteps, dt, dx, scheme; close(u)
  allocate(a(nx,nz,ny), au(nx+1,nz,ny), av(nx,nz,ny+1), acc(nx,ny))
  call hip_create(ctx, 0, 1, nx, 1, nz, 1, ny)
  do i = 1, 13
     call rd(trim(dir)//"/"//trim(n3(i))//".bin", a); call hip_upload(ctx, f3(i), a)
  end do
  call rdu(trim(dir)//"/u.bin", au); call hip_upload(ctx, ICAR_F_U, au)
  call rdu(trim(dir)//"/jacobian_u.bin", au); call hip_upload(ctx, ICAR_F_JACOBIAN_U, au)
  call rdu(trim(dir)//"/v.bin", av); call hip_upload(ctx, ICAR_F_V, av)
  call rdu(trim(dir)//"/jacobian_v.bin", av); call hip_upload(ctx, ICAR_F_JACOBIAN_V, av)
  if (scheme == 3) call hip_mpdata_exact(ctx, .true.)
  do s = 1, nsteps
     call hip_mp_simple(ctx, dt, 2, nx-1, 2, ny-1, 1, nz)             ! mp()   time_step.f90:512-523
     call hip_advect(ctx, min(scheme, 2), 2, .true., .false., dt, dx, adv)    ! advect time_step.f90:529 (kADV_UPWIND / kADV_MPDATA)
  end do
  do i = 9, 13
     call hip_download(ctx, f3(i), a); call wr(trim(dir)//"/out_"//trim(n3(i))//".bin", a)
  end do
  call hip_download_2dd(ctx, ICAR_F_PRECIPITATION, acc)
  open(newunit=u, file=trim(dir)//"/out_precip.bin", access="stream", form="unformatted", status="replace"); write(u) acc; close(u)
  call hip_destroy(ctx)
  print *, "icar_hip_demo: ok"
contains
  subroutine rd(fn, x)
    character(len=*), intent(in) :: fn
    real(c_float), intent(out) :: x(:,:,:)
    integer :: uu
    open(newunit=uu, file=fn, access="stream", form="unformatted", status="old"); read(uu) x; close(uu)
  end subroutine
  subroutine rdu(fn, x)
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
