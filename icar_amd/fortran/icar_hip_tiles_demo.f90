!> icar_hip_tiles_demo.f90 -- ONE IMAGE of a multi-image run driven from Fortran through the C ABI: the whole
!! step(domain, end_time, options) loop of time_step.f90:440-551 as one library call (hip_step), halos exchanged by the
!! library's own transport (hip_comm_init* + the halo_send / halo_retrieve inside the sub-step), dt reduced over the images
!! (co_min) inside hip_update_dt.  The reference starts N coarray images of one program; this image has no coarray runtime
!! (SURVEY 8c), so the test starts N OS processes of this program and passes the image number on the command line:
!!     icar_hip_tiles_demo <dir> <rank 0..N-1> <N> <shm name> [wind_iterations]
!! With the fifth argument update_winds(domain, options) runs first with windtype = kITERATIVE_WINDS (wind.f90:289-369,
!! :371-498: one library call, exchange_u / exchange_v per sweep over the same transport) and u, v, w are written out too.
!! With a coarray runtime the two lines marked (*) become this_image()-1 / num_images(), and the host-staged transport becomes
!!     if (this_image()==1) call hip_comm_unique_id(uid);  call co_broadcast(uid, 1);  call hip_comm_init(ctx, N, rank, uid, nb)
!! (RCCL over xGMI, one GPU per image; INTEGRATION.md section 4).  Reads the tile written by tests/test_gpu_fortran_host.py
!! (raw REAL(4), Fortran order, halos included) and writes the prognostic fields back.
program icar_hip_tiles_demo
  use iso_c_binding
  use icar_hip
  implicit none
  type(hip_ctx_t) :: ctx
  type(hip_step_config_t) :: cfg
  integer :: rank, nranks, u, i, nsteps, nz, wind_iterations
  integer :: ims, ime, jms, jme, its, ite, jts, jte, ids, ide, jds, jde, bnd(4)
  integer(c_int) :: nb(4)
  integer(c_size_t) :: slot_bytes
  real :: dx
  double precision :: end_time, clock
  real(c_float), allocatable, target :: a(:,:,:), au(:,:,:), av(:,:,:), dz_levels(:)
  real(c_double), allocatable, target :: acc(:,:), theta(:,:)
  character(len=512) :: dir, arg, shm
  integer, parameter :: n3 = 13
  integer(c_int), parameter :: f3(n3) = [ICAR_F_W, ICAR_F_PRESSURE, ICAR_F_EXNER, ICAR_F_DENSITY, ICAR_F_DZ_MASS, ICAR_F_JACOBIAN, &
       ICAR_F_JACOBIAN_W, ICAR_F_ADVECTION_DZ, ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, ICAR_F_POTENTIAL_TEMPERATURE]
  character(len=24), parameter :: names(n3) = [character(len=24) :: "w", "pressure", "exner", "density", "dz_mass", "jacobian", &
       "jacobian_w", "advection_dz", "water_vapor", "cloud_water", "rain", "snow", "potential_temperature"]
  integer(c_int), parameter :: adv(5) = [ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, ICAR_F_POTENTIAL_TEMPERATURE]

  call get_command_argument(1, dir)
  call get_command_argument(2, arg); read(arg,*) rank            ! (*) this_image() - 1
  call get_command_argument(3, arg); read(arg,*) nranks          ! (*) num_images()
  call get_command_argument(4, shm)
  wind_iterations = -1
  if (command_argument_count() >= 5) then
     call get_command_argument(5, arg); read(arg,*) wind_iterations
  end if
  open(newunit=u, file=trim(dir)//"/meta.txt", status="old")
  read(u,*) ims, ime, jms, jme, nz                               ! grid_t of this image (grid_obj.f90:39-255), global index space
  read(u,*) its, ite, jts, jte
  read(u,*) ids, ide, jds, jde
  read(u,*) nb                                                   ! north, south, east, west neighbour (0-based rank) or -1
  read(u,*) bnd                                                  ! west, east, south, north boundary flags
  read(u,*) end_time, dx, slot_bytes
  close(u)
  allocate(a(ims:ime,nz,jms:jme), au(ims:ime+1,nz,jms:jme), av(ims:ime,nz,jms:jme+1), acc(ims:ime,jms:jme), dz_levels(nz))
  open(newunit=u, file=trim(dir)//"/dz_levels.bin", access="stream", form="unformatted", status="old"); read(u) dz_levels; close(u)
  call hip_create(ctx, 0, ims, ime, 1, nz, jms, jme)
  do i = 1, n3
     call rd(trim(dir)//"/"//trim(names(i))//".bin", a); call hip_upload(ctx, f3(i), a)
  end do
  call rd(trim(dir)//"/u.bin", au);          call hip_upload(ctx, ICAR_F_U, au)
  call rd(trim(dir)//"/jacobian_u.bin", au); call hip_upload(ctx, ICAR_F_JACOBIAN_U, au)
  call rd(trim(dir)//"/v.bin", av);          call hip_upload(ctx, ICAR_F_V, av)
  call rd(trim(dir)//"/jacobian_v.bin", av); call hip_upload(ctx, ICAR_F_JACOBIAN_V, av)

  ! exchangeable%set_neighbors (exchangeable_obj.f90:69-115) + the transport
  if (nranks > 1) then
     call hip_comm_init_host(ctx, nranks, rank, trim(shm), slot_bytes, nb)
  else
     call hip_comm_init_local(ctx, nb)
  end if

  ! options_t / grid_t members the loop reads: upwind advection + mp_simple (configuration 1 of BASELINE.json)
  cfg%advection = 1; cfg%microphysics = 2; cfg%dx = dx; cfg%diagnostics = 0
  cfg%its = its; cfg%ite = ite; cfg%jts = jts; cfg%jte = jte; cfg%kts = 1; cfg%kte = nz
  cfg%ids = ids; cfg%ide = ide; cfg%jds = jds; cfg%jde = jde; cfg%kds = 1; cfg%kde = nz
  cfg%west_boundary = bnd(1); cfg%east_boundary = bnd(2); cfg%south_boundary = bnd(3); cfg%north_boundary = bnd(4)
  cfg%n_advect = 5; cfg%advect_fields(1:5) = adv
  cfg%n_exchange = 5; cfg%exchange_fields(1:5) = adv
  call hip_step_configure(ctx, cfg, dz_levels)
  if (wind_iterations >= 0) then
     ! init_winds (wind.f90:512-590) on an unrotated grid: sintheta = 0, costheta = 1; then update_winds (first call: on u, v, w)
     allocate(theta(ims:ime,jms:jme))
     theta = 0.0d0; call hip_upload_2dd(ctx, ICAR_F_SINTHETA, theta)
     theta = 1.0d0; call hip_upload_2dd(ctx, ICAR_F_COSTHETA, theta)
     call hip_update_winds(ctx, 3, wind_iterations, dx, 1)
     call hip_download(ctx, ICAR_F_U, au); call wr(trim(dir)//"/out_u.bin", au)
     call hip_download(ctx, ICAR_F_V, av); call wr(trim(dir)//"/out_v.bin", av)
     call hip_download(ctx, ICAR_F_W, a);  call wr(trim(dir)//"/out_w.bin", a)
  end if
  call hip_mp_reset(ctx)                                         ! mp_init
  call hip_set_model_time(ctx, 0.0d0)

  nsteps = hip_step(ctx, end_time)                               ! step(domain, end_time, options)

  do i = 9, n3
     call hip_download(ctx, f3(i), a); call wr(trim(dir)//"/out_"//trim(names(i))//".bin", a)
  end do
  call hip_download_2dd(ctx, ICAR_F_PRECIPITATION, acc)
  open(newunit=u, file=trim(dir)//"/out_precip.bin", access="stream", form="unformatted", status="replace"); write(u) acc; close(u)
  clock = hip_model_time(ctx)                                     ! domain%model_time%seconds()
  call hip_comm_destroy(ctx)
  call hip_destroy(ctx)
  print *, "icar_hip_tiles_demo: ok", nsteps, clock
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
