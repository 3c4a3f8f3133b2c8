!> icar_hip_mod.f90 -- Fortran 2008 host side of libicar_hip.so (include/icar_hip.h).
!!
!! `use icar_hip` gives (a) the raw iso_c_binding interfaces of every C-ABI entry point and
!! (b) thin wrappers with the reference's operator names and argument meaning:
!!     call hip_advect(ctx, options%physics%advection, mpdata_order, fct, advect_density, dt, dx, vars)
!!     call hip_mp_simple(ctx, dt, its,ite, jts,jte, kts,kte)
!!     call hip_thompson_init(ctx, mp_options...) ; call hip_thompson(ctx, dt, its..kte, ids..kde)
!! A non-zero return code becomes `error stop` with the library's message -- the reference's own
!! error convention (stop / error stop; SURVEY 8b).  INTEGRATION.md shows where ICAR calls these.
module icar_hip
  use iso_c_binding
  implicit none
  private
  public :: hip_ctx_t, hip_create, hip_destroy, hip_upload, hip_download, hip_upload_2dd, hip_download_2dd, &
            hip_advect, hip_mp_simple, hip_thompson_init, hip_thompson, hip_max_courant, hip_balance_uvw, hip_sync, &
            hip_lt_options_t, hip_setup_linwinds, hip_linwinds_build_lut, hip_spatial_winds, hip_iterative_winds, &
            hip_iterative_winds_correct_w, hip_iterative_winds_sweep, &
            hip_diagnostic_update, hip_diagnostic_update_parts, hip_dqdt_upload, hip_apply_forcing, hip_enforce_limits, hip_halo_count, hip_halo_pack, &
            hip_halo_unpack, hip_mp_simple_tiles, hip_halo_pack_dirs, hip_halo_unpack_dirs, hip_thompson_tiles, hip_mass_conservative_acceleration, hip_balance_uvw_update, hip_wsm3_init, hip_wsm3, hip_wsm6_init, hip_wsm6, hip_wsm6_tiles, hip_winds_valid, hip_max_courant_prefetch, &
            hip_aux_fork, hip_aux_begin, hip_aux_end, hip_aux_join, hip_max_courant_device, hip_make_winds_grid_relative, &
            hip_step_config_t, hip_step_configure, hip_update_dt, hip_compute_dt, hip_substep, hip_step, hip_step_n, hip_mp, hip_advect_step, hip_mp_reset, &
            hip_model_time, hip_set_model_time, hip_comm_unique_id, hip_comm_init, hip_comm_init_local, hip_comm_init_host, hip_comm_destroy, &
            hip_halo_send, hip_halo_retrieve, hip_co_min, hip_comm_ranks, hip_halo_selfcheck, hip_update_winds, hip_exchange_uv, hip_mpdata_exact, ICAR_NEIGHBOR_NONE, ICAR_NEIGHBOR_SELF, ICAR_N_ADVECTABLE
  public :: ICAR_F_WATER_VAPOR, ICAR_F_CLOUD_WATER, ICAR_F_RAIN, ICAR_F_SNOW, ICAR_F_POTENTIAL_TEMPERATURE, &
            ICAR_F_CLOUD_ICE, ICAR_F_GRAUPEL, ICAR_F_ICE_NUMBER, ICAR_F_RAIN_NUMBER, ICAR_F_U, ICAR_F_V, ICAR_F_W, &
            ICAR_F_PRESSURE, ICAR_F_EXNER, ICAR_F_DENSITY, ICAR_F_DZ_MASS, ICAR_F_JACOBIAN, ICAR_F_JACOBIAN_U, &
            ICAR_F_JACOBIAN_V, ICAR_F_JACOBIAN_W, ICAR_F_ADVECTION_DZ, ICAR_F_PRECIPITATION, ICAR_F_SNOWFALL, ICAR_F_GRAUPEL_ACC, &
            ICAR_F_Z, ICAR_F_NSQUARED, ICAR_F_PRESSURE_INTERFACE, ICAR_F_TEMPERATURE, ICAR_F_TEMPERATURE_INTERFACE, &
            ICAR_F_U_MASS, ICAR_F_V_MASS, ICAR_F_W_REAL, ICAR_F_DZDX, ICAR_F_DZDY, ICAR_F_SURFACE_PRESSURE, &
            ICAR_F_IVT, ICAR_F_IWV, ICAR_F_IWL, ICAR_F_IWI, ICAR_F_ZR_U, ICAR_F_ZR_V, ICAR_F_SINTHETA, ICAR_F_COSTHETA

  ! enum icar_hip_field (include/icar_hip.h)
  integer(c_int), parameter :: ICAR_F_WATER_VAPOR=0, ICAR_F_CLOUD_WATER=1, ICAR_F_RAIN=2, ICAR_F_SNOW=3, &
       ICAR_F_POTENTIAL_TEMPERATURE=4, ICAR_F_CLOUD_ICE=5, ICAR_F_GRAUPEL=6, ICAR_F_ICE_NUMBER=7, ICAR_F_RAIN_NUMBER=8, &
       ICAR_F_U=11, ICAR_F_V=12, ICAR_F_W=13, ICAR_F_PRESSURE=14, ICAR_F_EXNER=15, ICAR_F_DENSITY=16, ICAR_F_DZ_MASS=17, &
       ICAR_F_JACOBIAN=18, ICAR_F_JACOBIAN_U=19, ICAR_F_JACOBIAN_V=20, ICAR_F_JACOBIAN_W=21, ICAR_F_ADVECTION_DZ=22, &
       ICAR_F_PRECIPITATION=23, ICAR_F_SNOWFALL=24, ICAR_F_GRAUPEL_ACC=25, ICAR_F_PRESSURE_INTERFACE=26, ICAR_F_TEMPERATURE=27, &
       ICAR_F_TEMPERATURE_INTERFACE=28, ICAR_F_U_MASS=29, ICAR_F_V_MASS=30, ICAR_F_W_REAL=31, ICAR_F_DZDX=32, ICAR_F_DZDY=33, &
       ICAR_F_SURFACE_PRESSURE=34, ICAR_F_Z=35, ICAR_F_NSQUARED=36, ICAR_F_IVT=37, ICAR_F_IWV=38, ICAR_F_IWL=39, ICAR_F_IWI=40, &
       ICAR_F_ZR_U=41, ICAR_F_ZR_V=42, ICAR_F_SINTHETA=43, ICAR_F_COSTHETA=44

  integer(c_int), parameter :: ICAR_N_ADVECTABLE = 11, ICAR_NEIGHBOR_NONE = -1, ICAR_NEIGHBOR_SELF = -2

  !> struct icar_hip_step_config == the members of options_t / grid_t the sub-step loop reads (time_step.f90:440-551)
  type, bind(C) :: hip_step_config_t
     integer(c_int) :: advection = 0, microphysics = 0, mpdata_order = 2, flux_corrected_transport = 1, advect_density = 0
     integer(c_int) :: cfl_strictness = 3
     real(c_float)  :: cfl_reduction_factor = 0.9, dx = 1000.0, mp_update_interval = 0.0
     integer(c_int) :: top_mp_level = 0, halo_size = 1
     integer(c_int) :: its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde
     integer(c_int) :: west_boundary = 1, east_boundary = 1, south_boundary = 1, north_boundary = 1
     integer(c_int) :: diagnostics = 1, prefetch_dt = 1
     integer(c_int) :: n_advect = 0, advect_fields(ICAR_N_ADVECTABLE) = 0
     integer(c_int) :: n_exchange = 0, exchange_fields(ICAR_N_ADVECTABLE) = 0
     integer(c_int) :: n_forced = 0, forced_fields(16) = 0, force_boundaries(16) = 0
  end type

  !> struct icar_hip_lt_options == the members of options%lt_options the linear-wind path reads
  type, bind(C) :: hip_lt_options_t
     integer(c_int) :: buffer, stability_window_size, vert_smooth, variable_N, smooth_nsq
     real(c_float)  :: max_stability, min_stability, N_squared, linear_contribution, linear_update_fraction
     real(c_float)  :: dirmax, dirmin, spdmax, spdmin, nsqmax, nsqmin
     integer(c_int) :: n_dir_values, n_nsq_values, n_spd_values
     real(c_float)  :: minimum_layer_size
  end type

  type :: hip_ctx_t
     type(c_ptr) :: p = c_null_ptr
  end type

  interface
     integer(c_int) function icar_hip_ctx_create(ctx, device, ims, ime, kms, kme, jms, jme) bind(C, name="icar_hip_ctx_create")
       import; type(c_ptr), intent(out) :: ctx; integer(c_int), value :: device, ims, ime, kms, kme, jms, jme
     end function
     integer(c_int) function icar_hip_ctx_destroy(ctx) bind(C, name="icar_hip_ctx_destroy")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_synchronize(ctx) bind(C, name="icar_hip_synchronize")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_make_winds_grid_relative(ctx, update) bind(C, name="icar_hip_make_winds_grid_relative")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: update
     end function
     integer(c_int) function icar_hip_aux_fork(ctx) bind(C, name="icar_hip_aux_fork")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_aux_begin(ctx) bind(C, name="icar_hip_aux_begin")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_aux_end(ctx) bind(C, name="icar_hip_aux_end")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_aux_join(ctx) bind(C, name="icar_hip_aux_join")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_max_courant_device(ctx, dx, dz_levels, d_out) bind(C, name="icar_hip_max_courant_device")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx; real(c_float), intent(in) :: dz_levels(*); type(c_ptr), value :: d_out
     end function
     integer(c_int) function icar_hip_field_upload(ctx, field, host) bind(C, name="icar_hip_field_upload")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: field; type(c_ptr), value :: host
     end function
     integer(c_int) function icar_hip_field_download(ctx, field, host) bind(C, name="icar_hip_field_download")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: field; type(c_ptr), value :: host
     end function
     integer(c_int) function icar_hip_setup_winds(ctx, scheme, dt, dx, advect_density) bind(C, name="icar_hip_setup_winds")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: scheme, advect_density; real(c_float), value :: dt, dx
     end function
     integer(c_int) function icar_hip_advect(ctx, scheme, mpdata_order, fct, advect_density, fields, nfields) bind(C, name="icar_hip_advect")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: scheme, mpdata_order, fct, advect_density, nfields
       integer(c_int), intent(in) :: fields(*)
     end function
     integer(c_int) function icar_hip_mp_simple(ctx, dt, its, ite, jts, jte, kts, kte, err_count) bind(C, name="icar_hip_mp_simple")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt; integer(c_int), value :: its, ite, jts, jte, kts, kte
       type(c_ptr), value :: err_count
     end function
     integer(c_int) function icar_hip_thompson_init(ctx, params, flags) bind(C, name="icar_hip_thompson_init")
       import; type(c_ptr), value :: ctx; real(c_float), intent(in) :: params(18); integer(c_int), intent(in) :: flags(2)
     end function
     integer(c_int) function icar_hip_thompson(ctx, dt, its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde) bind(C, name="icar_hip_thompson")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt
       integer(c_int), value :: its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde
     end function
     integer(c_int) function icar_hip_max_courant(ctx, dx, dz_levels, res) bind(C, name="icar_hip_max_courant")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx; real(c_float), intent(in) :: dz_levels(*); real(c_float), intent(out) :: res
     end function
     integer(c_int) function icar_hip_balance_uvw(ctx, dx) bind(C, name="icar_hip_balance_uvw")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx
     end function
     integer(c_int) function icar_hip_max_courant_prefetch(ctx, dx, dz_levels) bind(C, name="icar_hip_max_courant_prefetch")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx; real(c_float), intent(in) :: dz_levels(*)
     end function
     integer(c_int) function icar_hip_winds_valid(ctx) bind(C, name="icar_hip_winds_valid")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_wsm6_init(ctx) bind(C, name="icar_hip_wsm6_init")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_wsm6(ctx, dt, its, ite, jts, jte, kts, kte) bind(C, name="icar_hip_wsm6")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt; integer(c_int), value :: its, ite, jts, jte, kts, kte
     end function
     integer(c_int) function icar_hip_wsm6_tiles(ctx, dt, ntiles, tiles, kts, kte) bind(C, name="icar_hip_wsm6_tiles")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt; integer(c_int), value :: ntiles, kts, kte
       integer(c_int), intent(in) :: tiles(4,*)
     end function
     integer(c_int) function icar_hip_wsm3_init(ctx) bind(C, name="icar_hip_wsm3_init")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_wsm3(ctx, dt, its, ite, jts, jte, kts, kte) bind(C, name="icar_hip_wsm3")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt; integer(c_int), value :: its, ite, jts, jte, kts, kte
     end function
     integer(c_int) function icar_hip_diagnostic_update_parts(ctx, parts) bind(C, name="icar_hip_diagnostic_update_parts")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: parts
     end function
     integer(c_int) function icar_hip_diagnostic_update(ctx) bind(C, name="icar_hip_diagnostic_update")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_dqdt_upload(ctx, field, host) bind(C, name="icar_hip_dqdt_upload")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: field; type(c_ptr), value :: host
     end function
     integer(c_int) function icar_hip_apply_forcing(ctx, dt, fields, fb, n, w, e, s, nn) bind(C, name="icar_hip_apply_forcing")
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt; integer(c_int), intent(in) :: fields(*), fb(*)
       integer(c_int), value :: n, w, e, s, nn
     end function
     integer(c_int) function icar_hip_enforce_limits(ctx, fields, n) bind(C, name="icar_hip_enforce_limits")
       import; type(c_ptr), value :: ctx; integer(c_int), intent(in) :: fields(*); integer(c_int), value :: n
     end function
     integer(c_size_t) function icar_hip_halo_count(ctx, dir, halo) bind(C, name="icar_hip_halo_count")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: dir, halo
     end function
     integer(c_int) function icar_hip_halo_pack(ctx, dir, halo, fields, n, dbuf) bind(C, name="icar_hip_halo_pack")
       import; type(c_ptr), value :: ctx, dbuf; integer(c_int), value :: dir, halo, n; integer(c_int), intent(in) :: fields(*)
     end function
     integer(c_int) function icar_hip_halo_unpack(ctx, dir, halo, fields, n, dbuf) bind(C, name="icar_hip_halo_unpack")
       import; type(c_ptr), value :: ctx, dbuf; integer(c_int), value :: dir, halo, n; integer(c_int), intent(in) :: fields(*)
     end function
     integer(c_int) function icar_hip_mp_simple_tiles(ctx, dt, ntiles, tiles, kts, kte, err_count) bind(C, name="icar_hip_mp_simple_tiles")
       import; type(c_ptr), value :: ctx, err_count; real(c_float), value :: dt; integer(c_int), value :: ntiles, kts, kte
       integer(c_int), intent(in) :: tiles(4,*)
     end function
     integer(c_int) function icar_hip_halo_pack_dirs(ctx, ndirs, dirs, halo, fields, n, dbufs) bind(C, name="icar_hip_halo_pack_dirs")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: ndirs, halo, n
       integer(c_int), intent(in) :: dirs(*), fields(*); type(c_ptr), intent(in) :: dbufs(*)
     end function
     integer(c_int) function icar_hip_halo_unpack_dirs(ctx, ndirs, dirs, halo, fields, n, dbufs) bind(C, name="icar_hip_halo_unpack_dirs")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: ndirs, halo, n
       integer(c_int), intent(in) :: dirs(*), fields(*); type(c_ptr), intent(in) :: dbufs(*)
     end function
     integer(c_int) function icar_hip_thompson_tiles(ctx, dt, ntiles, tiles, kts, kte, ids, ide, jds, jde, kds, kde) &
          bind(C, name="icar_hip_thompson_tiles")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dt; integer(c_int), value :: ntiles, kts, kte, ids, ide, jds, jde, kds, kde
       integer(c_int), intent(in) :: tiles(4,*)
     end function
     integer(c_int) function icar_hip_mass_conservative_acceleration(ctx, update) bind(C, name="icar_hip_mass_conservative_acceleration")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: update
     end function
     integer(c_int) function icar_hip_iterative_winds_correct_w(ctx, update) bind(C, name="icar_hip_iterative_winds_correct_w")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: update
     end function
     integer(c_int) function icar_hip_iterative_winds_sweep(ctx, dx, nsweeps, update) bind(C, name="icar_hip_iterative_winds_sweep")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx; integer(c_int), value :: nsweeps, update
     end function
     integer(c_int) function icar_hip_balance_uvw_update(ctx, dx) bind(C, name="icar_hip_balance_uvw_update")
       import; type(c_ptr), value :: ctx; real(c_float), value :: dx
     end function
     integer(c_int) function icar_hip_linwinds_setup(ctx, opt, terrain, nxg, nyg, ids, jds, dx) bind(C, name="icar_hip_linwinds_setup")
       import; type(c_ptr), value :: ctx; type(hip_lt_options_t), intent(in) :: opt; real(c_float), intent(in) :: terrain(*)
       integer(c_int), value :: nxg, nyg, ids, jds; real(c_float), value :: dx
     end function
     integer(c_int) function icar_hip_linwinds_build_lut(ctx, z_bottom, z_top, nz) bind(C, name="icar_hip_linwinds_build_lut")
       import; type(c_ptr), value :: ctx; real(c_float), intent(in) :: z_bottom(*), z_top(*); integer(c_int), value :: nz
     end function
     integer(c_int) function icar_hip_spatial_winds(ctx, update) bind(C, name="icar_hip_spatial_winds")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: update
     end function
     type(c_ptr) function icar_hip_last_error() bind(C, name="icar_hip_last_error")
       import
     end function
     ! ---- H1 transport + co_min, and the sub-step loop itself (comm.hip, timestep.hip) ----
     integer(c_int) function icar_hip_comm_unique_id(uid) bind(C, name="icar_hip_comm_unique_id")
       import; character(kind=c_char) :: uid(128)
     end function
     integer(c_int) function icar_hip_comm_init(ctx, nranks, rank, uid, neighbors) bind(C, name="icar_hip_comm_init")
       import; type(c_ptr), value :: ctx, uid; integer(c_int), value :: nranks, rank; integer(c_int), intent(in) :: neighbors(4)
     end function
     integer(c_int) function icar_hip_comm_init_host(ctx, nranks, rank, shm_name, slot_bytes, neighbors) bind(C, name="icar_hip_comm_init_host")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: nranks, rank; character(kind=c_char), intent(in) :: shm_name(*)
       integer(c_size_t), value :: slot_bytes; integer(c_int), intent(in) :: neighbors(4)
     end function
     integer(c_int) function icar_hip_comm_destroy(ctx) bind(C, name="icar_hip_comm_destroy")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_halo_send(ctx, halo, fields, n) bind(C, name="icar_hip_halo_send")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: halo, n; integer(c_int), intent(in) :: fields(*)
     end function
     integer(c_int) function icar_hip_halo_retrieve(ctx, halo, fields, n) bind(C, name="icar_hip_halo_retrieve")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: halo, n; integer(c_int), intent(in) :: fields(*)
     end function
     integer(c_int) function icar_hip_co_min(ctx, v) bind(C, name="icar_hip_co_min")
       import; type(c_ptr), value :: ctx; real(c_double), intent(inout) :: v
     end function
     integer(c_int) function icar_hip_update_winds(ctx, windtype, wind_iterations, dx, halo, update) bind(C, name="icar_hip_update_winds")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: windtype, wind_iterations, halo, update; real(c_float), value :: dx
     end function
     integer(c_int) function icar_hip_exchange_uv(ctx, halo, update) bind(C, name="icar_hip_exchange_uv")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: halo, update
     end function
     integer(c_int) function icar_hip_comm_ranks(ctx, nranks) bind(C, name="icar_hip_comm_ranks")
       import; type(c_ptr), value :: ctx; integer(c_int), intent(out) :: nranks
     end function
     integer(c_int) function icar_hip_mpdata_exact(ctx, on) bind(C, name="icar_hip_mpdata_exact")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: on
     end function
     integer(c_int) function icar_hip_halo_selfcheck(ctx, halo, n_bad) bind(C, name="icar_hip_halo_selfcheck")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: halo; integer(c_int), intent(out) :: n_bad
     end function
     integer(c_int) function icar_hip_step_configure(ctx, cfg, dz_levels) bind(C, name="icar_hip_step_configure")
       import; type(c_ptr), value :: ctx; type(hip_step_config_t), intent(in) :: cfg; real(c_float), intent(in) :: dz_levels(*)
     end function
     integer(c_int) function icar_hip_model_time_set(ctx, seconds) bind(C, name="icar_hip_model_time_set")
       import; type(c_ptr), value :: ctx; real(c_double), value :: seconds
     end function
     real(c_double) function icar_hip_model_time(ctx) bind(C, name="icar_hip_model_time")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_mp_reset(ctx) bind(C, name="icar_hip_mp_reset")
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function icar_hip_compute_dt(ctx, dt) bind(C, name="icar_hip_compute_dt")
       import; type(c_ptr), value :: ctx; real(c_double), intent(out) :: dt
     end function
     integer(c_int) function icar_hip_update_dt(ctx, dt) bind(C, name="icar_hip_update_dt")
       import; type(c_ptr), value :: ctx; real(c_double), intent(out) :: dt
     end function
     integer(c_int) function icar_hip_mp(ctx, dt, halo, subset) bind(C, name="icar_hip_mp")
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt; integer(c_int), value :: halo, subset
     end function
     integer(c_int) function icar_hip_advect_step(ctx, dt) bind(C, name="icar_hip_advect_step")
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt
     end function
     integer(c_int) function icar_hip_substep(ctx, dt, enforce) bind(C, name="icar_hip_substep")
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt; integer(c_int), value :: enforce
     end function
     integer(c_int) function icar_hip_step(ctx, end_time, nsteps) bind(C, name="icar_hip_step")
       import; type(c_ptr), value :: ctx; real(c_double), value :: end_time; integer(c_int), intent(out) :: nsteps
     end function
     integer(c_int) function icar_hip_step_n(ctx, nsteps, dt_last) bind(C, name="icar_hip_step_n")
       import; type(c_ptr), value :: ctx; integer(c_int), value :: nsteps; real(c_double), intent(out) :: dt_last
     end function
  end interface

contains

  !> hand the library the options_t / grid_t members the loop reads; then hip_step == step(domain, end_time, options)
  subroutine hip_step_configure(ctx, cfg, dz_levels)
    type(hip_ctx_t), intent(in) :: ctx
    type(hip_step_config_t), intent(in) :: cfg
    real(c_float), intent(in) :: dz_levels(:)
    call check(icar_hip_step_configure(ctx%p, cfg, dz_levels), "step_configure")
  end subroutine

  !> update_dt (time_step.f90:375-423): compute_dt + co_min over the images + the 120 s cap
  function hip_update_dt(ctx) result(dt)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double) :: dt
    call check(icar_hip_update_dt(ctx%p, dt), "update_dt")
  end function

  function hip_compute_dt(ctx) result(dt)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double) :: dt
    call check(icar_hip_compute_dt(ctx%p, dt), "compute_dt")
  end function

  !> one pass of time_step.f90:474-539 (diagnostic_update ... enforce_limits), two streams inside the library
  subroutine hip_substep(ctx, dt, enforce_limits)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(in) :: dt
    logical, intent(in) :: enforce_limits
    call check(icar_hip_substep(ctx%p, dt, merge(1_c_int,0_c_int,enforce_limits)), "substep")
  end subroutine

  !> step(domain, end_time, options) (time_step.f90:440-551); the model clock lives in the context (hip_model_time)
  function hip_step(ctx, end_time) result(nsteps)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(in) :: end_time
    integer :: nsteps
    integer(c_int) :: n
    call check(icar_hip_step(ctx%p, end_time, n), "step"); nsteps = n
  end function

  !> nsteps sub-steps (update_dt -> substep -> clock += dt) without an end time; returns the last dt
  function hip_step_n(ctx, nsteps) result(dt)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: nsteps
    real(c_double) :: dt
    call check(icar_hip_step_n(ctx%p, int(nsteps,c_int), dt), "step_n")
  end function

  !> mp(domain, options, dt, halo, subset) (mp_driver.f90:673-772); absent halo / subset like the reference's optionals
  subroutine hip_mp(ctx, dt, halo, subset)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(in) :: dt
    integer, intent(in), optional :: halo, subset
    integer(c_int) :: h, s
    h = -1; s = -1
    if (present(halo)) h = halo
    if (present(subset)) s = subset
    call check(icar_hip_mp(ctx%p, dt, h, s), "mp")
  end subroutine

  !> advect(domain, options, dt) (advection_driver.f90:51-77)
  subroutine hip_advect_step(ctx, dt)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(in) :: dt
    call check(icar_hip_advect_step(ctx%p, dt), "advect_step")
  end subroutine

  subroutine hip_mp_reset(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_mp_reset(ctx%p), "mp_reset")
  end subroutine

  function hip_model_time(ctx) result(t)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double) :: t
    t = icar_hip_model_time(ctx%p)
  end function

  subroutine hip_set_model_time(ctx, seconds)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(in) :: seconds
    call check(icar_hip_model_time_set(ctx%p, seconds), "model_time_set")
  end subroutine

  !> image 1 calls this and co_broadcasts the 128 bytes; every image then calls hip_comm_init (RCCL, one GPU per image)
  subroutine hip_comm_unique_id(uid)
    character(kind=c_char), intent(out) :: uid(128)
    call check(icar_hip_comm_unique_id(uid), "comm_unique_id")
  end subroutine

  !> neighbors(1:4) = 0-based rank (= image - 1) of the north, south, east, west neighbour, ICAR_NEIGHBOR_NONE on a domain boundary
  subroutine hip_comm_init(ctx, nranks, rank, uid, neighbors)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: nranks, rank
    character(kind=c_char), intent(in), target :: uid(128)
    integer(c_int), intent(in) :: neighbors(4)
    call check(icar_hip_comm_init(ctx%p, int(nranks,c_int), int(rank,c_int), c_loc(uid), neighbors), "comm_init")
  end subroutine

  !> one image without peers (edges may wrap: ICAR_NEIGHBOR_SELF)
  subroutine hip_comm_init_local(ctx, neighbors)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: neighbors(4)
    call check(icar_hip_comm_init(ctx%p, 1_c_int, 0_c_int, c_null_ptr, neighbors), "comm_init")
  end subroutine

  !> the same entry points staged through POSIX shared memory: several images on one GPU (functional path)
  subroutine hip_comm_init_host(ctx, nranks, rank, shm_name, slot_bytes, neighbors)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: nranks, rank
    character(len=*), intent(in) :: shm_name
    integer(c_size_t), intent(in) :: slot_bytes
    integer(c_int), intent(in) :: neighbors(4)
    call check(icar_hip_comm_init_host(ctx%p, int(nranks,c_int), int(rank,c_int), trim(shm_name)//c_null_char, slot_bytes, neighbors), "comm_init_host")
  end subroutine

  subroutine hip_comm_destroy(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_comm_destroy(ctx%p), "comm_destroy")
  end subroutine

  !> domain%halo_send / halo_retrieve (domain_obj.f90:109-143) for the listed exchangeables, one message per neighbour
  subroutine hip_halo_send(ctx, halo, fields)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: halo
    integer(c_int), intent(in) :: fields(:)
    call check(icar_hip_halo_send(ctx%p, int(halo,c_int), fields, int(size(fields),c_int)), "halo_send")
  end subroutine
  subroutine hip_halo_retrieve(ctx, halo, fields)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: halo
    integer(c_int), intent(in) :: fields(:)
    call check(icar_hip_halo_retrieve(ctx%p, int(halo,c_int), fields, int(size(fields),c_int)), "halo_retrieve")
  end subroutine

  !> call co_min(seconds) (time_step.f90:413)
  subroutine hip_co_min(ctx, v)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_double), intent(inout) :: v
    call check(icar_hip_co_min(ctx%p, v), "co_min")
  end subroutine

  !> num_images() as the transport reports it (ncclCommCount / the shared segment's header)
  integer function hip_comm_ranks(ctx) result(n)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int) :: nr
    call check(icar_hip_comm_ranks(ctx%p, nr), "comm_ranks")
    n = int(nr)
  end function

  !> one exchange of a rank-stamped field, verified on the device (exchangeable_obj.f90:138-356); returns the number of halo
  !! cells that do not carry their neighbour's stamp.  Collective.
  integer function hip_halo_selfcheck(ctx, halo) result(n_bad)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: halo
    integer(c_int) :: nb
    call check(icar_hip_halo_selfcheck(ctx%p, int(halo,c_int), nb), "halo_selfcheck")
    n_bad = int(nb)
  end function

  !> MPDATA's corrective iterations in the operation order of adv_mpdata.f90 (bit-identical to the CPU reference, ~4x the
  !! advection time of the fused kernel); .false. = the fused kernel (default)
  subroutine hip_mpdata_exact(ctx, on)
    type(hip_ctx_t), intent(in) :: ctx
    logical, intent(in) :: on
    call check(icar_hip_mpdata_exact(ctx%p, merge(1_c_int, 0_c_int, on)), "mpdata_exact")
  end subroutine

  subroutine check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (rc == 0) return
    call c_f_pointer(icar_hip_last_error(), msg, [512])
    n = 0
    do while (n < 512)
       if (msg(n+1) == c_null_char) exit
       n = n + 1
    end do
    write(*,*) "icar_hip: ", what, ": ", msg(1:n)
    error stop "icar_hip call failed"
  end subroutine

  !> one context per image; ims..jme are the tile's memory bounds (domain%grid%ims ...)
  subroutine hip_create(ctx, device, ims, ime, kms, kme, jms, jme)
    type(hip_ctx_t), intent(out) :: ctx
    integer, intent(in) :: device, ims, ime, kms, kme, jms, jme
    call check(icar_hip_ctx_create(ctx%p, int(device,c_int), int(ims,c_int), int(ime,c_int), int(kms,c_int), &
                                   int(kme,c_int), int(jms,c_int), int(jme,c_int)), "ctx_create")
  end subroutine

  subroutine hip_destroy(ctx)
    type(hip_ctx_t), intent(inout) :: ctx
    call check(icar_hip_ctx_destroy(ctx%p), "ctx_destroy"); ctx%p = c_null_ptr
  end subroutine

  subroutine hip_sync(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_synchronize(ctx%p), "synchronize")
  end subroutine

  !> make_winds_grid_relative(u, v, w, sintheta, costheta) of wind.f90:236-287 on the device u, v (update: their dqdt_3d);
  !! upload domain%sintheta / costheta with hip_upload_2dd(ctx, ICAR_F_SINTHETA / ICAR_F_COSTHETA, ...) once after init_winds
  subroutine hip_make_winds_grid_relative(ctx, update)
    type(hip_ctx_t), intent(in) :: ctx
    logical, intent(in) :: update
    call check(icar_hip_make_winds_grid_relative(ctx%p, merge(1_c_int,0_c_int,update)), "make_winds_grid_relative")
  end subroutine

  !> second HIP stream: `call hip_aux_fork(ctx)` before mp(halo=1); the interior mp(subset=1) between
  !! hip_aux_begin / hip_aux_end runs beside the strips + halo_send; hip_aux_join before halo_retrieve (time_step.f90:512-526)
  subroutine hip_aux_fork(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_aux_fork(ctx%p), "aux_fork")
  end subroutine
  subroutine hip_aux_begin(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_aux_begin(ctx%p), "aux_begin")
  end subroutine
  subroutine hip_aux_end(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_aux_end(ctx%p), "aux_end")
  end subroutine
  subroutine hip_aux_join(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_aux_join(ctx%p), "aux_join")
  end subroutine

  !> compute_dt's strictness-3 reduction left on the device (d_out = device address of one REAL(4)), for a device-side co_min
  subroutine hip_max_courant_device(ctx, dx, dz_levels, d_out)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx, dz_levels(:)
    type(c_ptr), intent(in) :: d_out
    call check(icar_hip_max_courant_device(ctx%p, real(dx,c_float), dz_levels, d_out), "max_courant_device")
  end subroutine

  !> domain%X%data_3d -> device.  The reference allocates these whole-array, i.e. contiguous; the
  !! pointers are not declared CONTIGUOUS (exchangeable_h.f90:14), so assert it before c_loc.
  subroutine hip_upload(ctx, field, a)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: field
    real(c_float), intent(in), target, contiguous :: a(:,:,:)
    call check(icar_hip_field_upload(ctx%p, field, c_loc(a)), "field_upload")
  end subroutine

  subroutine hip_download(ctx, field, a)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: field
    real(c_float), intent(inout), target, contiguous :: a(:,:,:)
    call check(icar_hip_field_download(ctx%p, field, c_loc(a)), "field_download")
  end subroutine

  subroutine hip_upload_2dd(ctx, field, a)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: field
    real(c_double), intent(in), target, contiguous :: a(:,:)
    call check(icar_hip_field_upload(ctx%p, field, c_loc(a)), "field_upload")
  end subroutine

  subroutine hip_download_2dd(ctx, field, a)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: field
    real(c_double), intent(inout), target, contiguous :: a(:,:)
    call check(icar_hip_field_download(ctx%p, field, c_loc(a)), "field_download")
  end subroutine

  !> advect(domain, options, dt) of advection_driver.f90:51 : scheme = options%physics%advection,
  !! vars = field ids with options%vars_to_advect(...)>0 in the order of adv_mpdata.f90:512-522.
  subroutine hip_advect(ctx, scheme, mpdata_order, fct, advect_density, dt, dx, vars)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: scheme, mpdata_order
    logical, intent(in) :: fct, advect_density
    real, intent(in) :: dt, dx
    integer(c_int), intent(in) :: vars(:)
    call check(icar_hip_setup_winds(ctx%p, int(scheme,c_int), real(dt,c_float), real(dx,c_float), &
                                    merge(1_c_int,0_c_int,advect_density)), "setup_winds")
    call check(icar_hip_advect(ctx%p, int(scheme,c_int), int(mpdata_order,c_int), merge(1_c_int,0_c_int,fct), &
                               merge(1_c_int,0_c_int,advect_density), vars, int(size(vars),c_int)), "advect")
  end subroutine

  !> mp_simple_driver (mp_simple.f90:595) on a tile, incl. the precipitation accumulation of process_subdomain
  subroutine hip_mp_simple(ctx, dt, its, ite, jts, jte, kts, kte)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer, intent(in) :: its, ite, jts, jte, kts, kte
    call check(icar_hip_mp_simple(ctx%p, real(dt,c_float), int(its,c_int), int(ite,c_int), int(jts,c_int), int(jte,c_int), &
                                  int(kts,c_int), int(kte,c_int), c_null_ptr), "mp_simple")
  end subroutine

  subroutine hip_thompson_init(ctx, params, Ef_rw_l, Ef_sw_l)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_float), intent(in) :: params(18)      ! mp_options_type in declaration order (opt_types.f90:30-41)
    logical, intent(in) :: Ef_rw_l, Ef_sw_l
    integer(c_int) :: flags(2)
    flags = [merge(1_c_int,0_c_int,Ef_rw_l), merge(1_c_int,0_c_int,Ef_sw_l)]
    call check(icar_hip_thompson_init(ctx%p, params, flags), "thompson_init")
  end subroutine

  subroutine hip_thompson(ctx, dt, its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer, intent(in) :: its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde
    call check(icar_hip_thompson(ctx%p, real(dt,c_float), int(its,c_int), int(ite,c_int), int(jts,c_int), int(jte,c_int), &
               int(kts,c_int), int(kte,c_int), int(ids,c_int), int(ide,c_int), int(jds,c_int), int(jde,c_int), &
               int(kds,c_int), int(kde,c_int)), "thompson")
  end subroutine

  !> the reduction inside compute_dt (time_step.f90:264-289); dt = CFL / result
  real function hip_max_courant(ctx, dx, dz_levels)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    real(c_float), intent(in) :: dz_levels(:)
    real(c_float) :: r
    call check(icar_hip_max_courant(ctx%p, real(dx,c_float), dz_levels, r), "max_courant")
    hip_max_courant = r
  end function

  subroutine hip_balance_uvw(ctx, dx)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    call check(icar_hip_balance_uvw(ctx%p, real(dx,c_float)), "balance_uvw")
  end subroutine

  !> the CFL reduction of the NEXT update_dt taken now, on the current stream (beside the advection, after the forcing of u, v, w)
  subroutine hip_max_courant_prefetch(ctx, dx, dz_levels)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    real(c_float), intent(in) :: dz_levels(:)
    call check(icar_hip_max_courant_prefetch(ctx%p, real(dx,c_float), dz_levels), "max_courant_prefetch")
  end subroutine

  !> process_halo's strips for WSM6 in one sequence of launches: tiles(1:4, t) = its, ite, jts, jte of strip t
  subroutine hip_wsm6_tiles(ctx, dt, tiles, kts, kte)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer(c_int), intent(in) :: tiles(:,:)
    integer, intent(in) :: kts, kte
    call check(icar_hip_wsm6_tiles(ctx%p, real(dt,c_float), int(size(tiles,2),c_int), tiles, int(kts,c_int), int(kte,c_int)), "wsm6_tiles")
  end subroutine

  !> .true. while the Courant winds of the last hip_setup_winds still belong to the state
  logical function hip_winds_valid(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    hip_winds_valid = icar_hip_winds_valid(ctx%p) /= 0
  end function

  !> wsm6init / wsm6 as mp_driver.f90:100 and :518-550 call them
  subroutine hip_wsm6_init(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_wsm6_init(ctx%p), "wsm6_init")
  end subroutine

  subroutine hip_wsm6(ctx, dt, its, ite, jts, jte, kts, kte)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer, intent(in) :: its, ite, jts, jte, kts, kte
    call check(icar_hip_wsm6(ctx%p, real(dt,c_float), int(its,c_int), int(ite,c_int), int(jts,c_int), int(jte,c_int), &
                             int(kts,c_int), int(kte,c_int)), "wsm6")
  end subroutine

  !> wsm3init / wsm3 as mp_driver.f90:105 and :552-585 call them (qci = cloud_water_mass, qrs = rain_mass, w = w_real)
  subroutine hip_wsm3_init(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_wsm3_init(ctx%p), "wsm3_init")
  end subroutine

  subroutine hip_wsm3(ctx, dt, its, ite, jts, jte, kts, kte)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer, intent(in) :: its, ite, jts, jte, kts, kte
    call check(icar_hip_wsm3(ctx%p, real(dt,c_float), int(its,c_int), int(ite,c_int), int(jts,c_int), int(jte,c_int), &
                             int(kts,c_int), int(kte,c_int)), "wsm3")
  end subroutine

  !> diagnostic_update in parts: 1 = all but w_real, 2 = w_real only (can run beside the interior microphysics), 3 = both
  subroutine hip_diagnostic_update_parts(ctx, parts)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: parts
    call check(icar_hip_diagnostic_update_parts(ctx%p, int(parts,c_int)), "diagnostic_update_parts")
  end subroutine

  !> diagnostic_update (time_step.f90:49-198)
  subroutine hip_diagnostic_update(ctx)
    type(hip_ctx_t), intent(in) :: ctx
    call check(icar_hip_diagnostic_update(ctx%p), "diagnostic_update")
  end subroutine

  !> variable%meta_data%dqdt_3d -> device (the forcing tendency apply_forcing adds)
  subroutine hip_dqdt_upload(ctx, field, a)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: field
    real(c_float), intent(in), target, contiguous :: a(:,:,:)
    call check(icar_hip_dqdt_upload(ctx%p, field, c_loc(a)), "dqdt_upload")
  end subroutine

  !> domain%apply_forcing(dt) (domain_obj.f90:2383-2448): force_boundaries(i) -> only the true domain edges
  subroutine hip_apply_forcing(ctx, dt_seconds, fields, force_boundaries, west, east, south, north)
    type(hip_ctx_t), intent(in) :: ctx
    double precision, intent(in) :: dt_seconds
    integer(c_int), intent(in) :: fields(:)
    logical, intent(in) :: force_boundaries(:), west, east, south, north
    integer(c_int) :: fb(size(fields))
    fb = merge(1_c_int, 0_c_int, force_boundaries)
    call check(icar_hip_apply_forcing(ctx%p, real(dt_seconds,c_double), fields, fb, int(size(fields),c_int), &
               merge(1_c_int,0_c_int,west), merge(1_c_int,0_c_int,east), merge(1_c_int,0_c_int,south), merge(1_c_int,0_c_int,north)), &
               "apply_forcing")
  end subroutine

  !> domain%enforce_limits() (domain_obj.f90:2228-2243)
  subroutine hip_enforce_limits(ctx, fields)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: fields(:)
    call check(icar_hip_enforce_limits(ctx%p, fields, int(size(fields),c_int)), "enforce_limits")
  end subroutine

  !> halo faces of all exchanged scalars in one device buffer per neighbour (exchangeable_obj.f90:248-356).
  !! dir: 0 north, 1 south, 2 east, 3 west.  dbuf is a DEVICE pointer (hipMalloc / GPU-aware MPI window).
  integer(c_size_t) function hip_halo_count(ctx, dir, halo)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: dir, halo
    hip_halo_count = icar_hip_halo_count(ctx%p, int(dir,c_int), int(halo,c_int))
  end function

  subroutine hip_halo_pack(ctx, dir, halo, fields, dbuf)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: dir, halo
    integer(c_int), intent(in) :: fields(:)
    type(c_ptr), intent(in) :: dbuf
    call check(icar_hip_halo_pack(ctx%p, int(dir,c_int), int(halo,c_int), fields, int(size(fields),c_int), dbuf), "halo_pack")
  end subroutine

  subroutine hip_halo_unpack(ctx, dir, halo, fields, dbuf)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: dir, halo
    integer(c_int), intent(in) :: fields(:)
    type(c_ptr), intent(in) :: dbuf
    call check(icar_hip_halo_unpack(ctx%p, int(dir,c_int), int(halo,c_int), fields, int(size(fields),c_int), dbuf), "halo_unpack")
  end subroutine

  !> process_halo's strips for mp_simple in one launch: tiles(1:4, t) = its, ite, jts, jte of strip t
  subroutine hip_mp_simple_tiles(ctx, dt, tiles, kts, kte)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer(c_int), intent(in) :: tiles(:,:)
    integer, intent(in) :: kts, kte
    call check(icar_hip_mp_simple_tiles(ctx%p, real(dt,c_float), int(size(tiles,2),c_int), tiles, int(kts,c_int), int(kte,c_int), &
                                        c_null_ptr), "mp_simple_tiles")
  end subroutine

  !> every direction of one halo_send / halo_retrieve in one launch (dirs(t), dbufs(t))
  subroutine hip_halo_pack_dirs(ctx, dirs, halo, fields, dbufs)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: dirs(:), fields(:)
    integer, intent(in) :: halo
    type(c_ptr), intent(in) :: dbufs(:)
    call check(icar_hip_halo_pack_dirs(ctx%p, int(size(dirs),c_int), dirs, int(halo,c_int), fields, int(size(fields),c_int), dbufs), "halo_pack_dirs")
  end subroutine

  subroutine hip_halo_unpack_dirs(ctx, dirs, halo, fields, dbufs)
    type(hip_ctx_t), intent(in) :: ctx
    integer(c_int), intent(in) :: dirs(:), fields(:)
    integer, intent(in) :: halo
    type(c_ptr), intent(in) :: dbufs(:)
    call check(icar_hip_halo_unpack_dirs(ctx%p, int(size(dirs),c_int), dirs, int(halo,c_int), fields, int(size(fields),c_int), dbufs), "halo_unpack_dirs")
  end subroutine

  !> process_halo's strips (mp_driver.f90:609-658) in one launch: tiles(1:4, t) = its, ite, jts, jte of strip t
  subroutine hip_thompson_tiles(ctx, dt, tiles, kts, kte, ids, ide, jds, jde, kds, kde)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dt
    integer(c_int), intent(in) :: tiles(:,:)
    integer, intent(in) :: kts, kte, ids, ide, jds, jde, kds, kde
    call check(icar_hip_thompson_tiles(ctx%p, real(dt,c_float), int(size(tiles,2),c_int), tiles, int(kts,c_int), int(kte,c_int), &
               int(ids,c_int), int(ide,c_int), int(jds,c_int), int(jde,c_int), int(kds,c_int), int(kde,c_int)), "thompson_tiles")
  end subroutine

  !> mass_conservative_acceleration (wind.f90:500-511) with ICAR_F_ZR_U / ICAR_F_ZR_V uploaded
  subroutine hip_mass_conservative_acceleration(ctx, update)
    type(hip_ctx_t), intent(in) :: ctx
    logical, intent(in) :: update
    call check(icar_hip_mass_conservative_acceleration(ctx%p, merge(1_c_int,0_c_int,update)), "mass_conservative_acceleration")
  end subroutine

  subroutine hip_balance_uvw_update(ctx, dx)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    call check(icar_hip_balance_uvw_update(ctx%p, real(dx,c_float)), "balance_uvw_update")
  end subroutine

  !> iterative_winds (wind.f90:371-498) on ONE image (exchange_u / exchange_v are no-ops there): balance_uvw, the model-top
  !> correction of w, then wind_iterations+1 Jacobi sweeps.  A multi-image host calls the two entry points itself and
  !> keeps its `call domain%u%exchange_u(); call domain%v%exchange_v()` between single sweeps.
  subroutine hip_iterative_winds(ctx, dx, wind_iterations, update)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    integer, intent(in) :: wind_iterations
    logical, intent(in), optional :: update
    integer(c_int) :: upd
    upd = 0
    if (present(update)) upd = merge(1, 0, update)
    if (upd == 1) then
       call check(icar_hip_balance_uvw_update(ctx%p, real(dx,c_float)), "balance_uvw_update")
    else
       call check(icar_hip_balance_uvw(ctx%p, real(dx,c_float)), "balance_uvw")
    end if
    call check(icar_hip_iterative_winds_correct_w(ctx%p, upd), "iterative_winds_correct_w")
    call check(icar_hip_iterative_winds_sweep(ctx%p, real(dx,c_float), int(wind_iterations+1,c_int), upd), "iterative_winds_sweep")
  end subroutine

  !> the two pieces of iterative_winds for a host that keeps the loop (and its exchange_u / exchange_v per sweep) in its own hands:
  !! the model-top correction of w (wind.f90:430-441) and nsweeps Jacobi sweeps (:455-481: calc_divergence, ADJ, the u / v faces)
  subroutine hip_iterative_winds_correct_w(ctx, update)
    type(hip_ctx_t), intent(in) :: ctx
    logical, intent(in) :: update
    call check(icar_hip_iterative_winds_correct_w(ctx%p, merge(1_c_int, 0_c_int, update)), "iterative_winds_correct_w")
  end subroutine
  subroutine hip_iterative_winds_sweep(ctx, dx, nsweeps, update)
    type(hip_ctx_t), intent(in) :: ctx
    real, intent(in) :: dx
    integer, intent(in) :: nsweeps
    logical, intent(in) :: update
    call check(icar_hip_iterative_winds_sweep(ctx%p, real(dx,c_float), int(nsweeps,c_int), merge(1_c_int, 0_c_int, update)), "iterative_winds_sweep")
  end subroutine

  !> update_winds(domain, options) (wind.f90:289-369) as one call, on any number of images: make_winds_grid_relative ->
  !! linear_perturb (windtype 1, 5) -> mass_conservative_acceleration (2) -> iterative_winds with its exchange_u / exchange_v
  !! per sweep (3, 5) -> balance_uvw.  The first call of a context works on u, v, w, later ones on their dqdt_3d, like the
  !! reference (`update` forces either).  windtype = options%physics%windtype, halo = grid%halo_size.
  subroutine hip_update_winds(ctx, windtype, wind_iterations, dx, halo, update)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: windtype, wind_iterations, halo
    real, intent(in) :: dx
    logical, intent(in), optional :: update
    integer(c_int) :: upd
    upd = -1
    if (present(update)) upd = merge(1, 0, update)
    call check(icar_hip_update_winds(ctx%p, int(windtype,c_int), int(wind_iterations,c_int), real(dx,c_float), int(halo,c_int), upd), "update_winds")
  end subroutine

  !> call domain%u%exchange_u(); call domain%v%exchange_v() (exchangeable_obj.f90:158-229), one message per neighbour
  subroutine hip_exchange_uv(ctx, halo, update)
    type(hip_ctx_t), intent(in) :: ctx
    integer, intent(in) :: halo
    logical, intent(in), optional :: update
    integer(c_int) :: upd
    upd = 0
    if (present(update)) upd = merge(1, 0, update)
    call check(icar_hip_exchange_uv(ctx%p, int(halo,c_int), upd), "exchange_uv")
  end subroutine

  !> setup_linwinds (linear_winds.f90:1180): terrain spectrum, wavenumber axes, zeroed perturbation state
  subroutine hip_setup_linwinds(ctx, lt, global_terrain, ids, jds, dx)
    type(hip_ctx_t), intent(in) :: ctx
    type(hip_lt_options_t), intent(in) :: lt
    real(c_float), contiguous, intent(in) :: global_terrain(:,:)
    integer, intent(in) :: ids, jds
    real, intent(in) :: dx
    call check(icar_hip_linwinds_setup(ctx%p, lt, global_terrain, int(size(global_terrain,1),c_int), &
               int(size(global_terrain,2),c_int), int(ids,c_int), int(jds,c_int), real(dx,c_float)), "linwinds_setup")
  end subroutine

  !> initialize_spatial_winds (linear_winds.f90:596), constant-z layers: z_bottom/top = layer_height -/+ dz_levels/2
  subroutine hip_linwinds_build_lut(ctx, z_bottom, z_top)
    type(hip_ctx_t), intent(in) :: ctx
    real(c_float), intent(in) :: z_bottom(:), z_top(:)
    call check(icar_hip_linwinds_build_lut(ctx%p, z_bottom, z_top, int(size(z_bottom),c_int)), "linwinds_build_lut")
  end subroutine

  !> spatial_winds (linear_winds.f90:840): update=.true. targets u/v dqdt_3d
  subroutine hip_spatial_winds(ctx, update)
    type(hip_ctx_t), intent(in) :: ctx
    logical, intent(in) :: update
    call check(icar_hip_spatial_winds(ctx%p, merge(1_c_int, 0_c_int, update)), "spatial_winds")
  end subroutine
end module icar_hip
