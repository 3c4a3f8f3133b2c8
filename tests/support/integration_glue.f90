!> tests/support/integration_glue.f90 -- TEST INFRASTRUCTURE (build container only; nothing of this travels to the GPU box).
!! The snippets of INTEGRATION.md sections 2, 3a, 3, 3b and 4, verbatim, inside subroutines whose dummies are the REFERENCE's own
!! domain_t / options_t (module files of /root/reference/src/objects/{domain_h,options_h,grid_h,opt_types}.f90 as compiled by
!! oracle/build_ref.sh into oracle/_ref/obj) next to `use icar_hip`: tests/test_integration_compiles.py runs flang -fsyntax-only
!! on it, so a member that INTEGRATION.md names and the reference does not have, or an argument of the wrong type / kind / rank, is
!! a compile error here (SURVEY section 8 row D1).  Lines marked "! [INTEGRATION section N]" are the documented ones; the rest is the
!! declarations a maintainer's surrounding routine already has.
module integration_glue
  use, intrinsic :: iso_c_binding
  use domain_interface,  only : domain_t
  use options_interface, only : options_t
  use options_types,     only : mp_options_type, lt_options_type
  use grid_interface,    only : grid_t
  use icar_constants,    only : kVARS, kMP_THOMPSON, kMP_SB04, kMP_WSM3, kMP_WSM6, kADV_MPDATA
  use time_object,       only : Time_type
  use time_delta_object, only : time_delta_t
  use icar_hip
  implicit none
contains

  !> section 2: one context per image, device mirrors of the domain_t members (objects/domain_obj.f90, end of domain%init)
  subroutine glue_init(this, hip, n_gpus_per_node, image)
    type(domain_t), intent(inout) :: this
    type(hip_ctx_t), intent(out) :: hip                                   ! (INTEGRATION: a member of domain_t, this%hip)
    integer, intent(in) :: n_gpus_per_node, image                         ! image = this_image()
    call hip_create(hip, device=mod(image-1, n_gpus_per_node), &
                    ims=this%grid%ims, ime=this%grid%ime, kms=this%grid%kms, kme=this%grid%kme, &
                    jms=this%grid%jms, jme=this%grid%jme)                                                 ! [INTEGRATION section 2]
    call hip_upload(hip, ICAR_F_JACOBIAN,     this%jacobian)                                              ! [INTEGRATION section 2]
    call hip_upload(hip, ICAR_F_JACOBIAN_U,   this%jacobian_u)
    call hip_upload(hip, ICAR_F_JACOBIAN_V,   this%jacobian_v)
    call hip_upload(hip, ICAR_F_JACOBIAN_W,   this%jacobian_w)
    call hip_upload(hip, ICAR_F_ADVECTION_DZ, this%advection_dz)
    call hip_upload(hip, ICAR_F_DZ_MASS,      this%dz_mass%data_3d)
    ! the synchronisation points of main/driver.f90:132-137 / :179,190
    call hip_upload(hip, ICAR_F_WATER_VAPOR,           this%water_vapor%data_3d)
    call hip_upload(hip, ICAR_F_POTENTIAL_TEMPERATURE, this%potential_temperature%data_3d)
    call hip_upload(hip, ICAR_F_U, this%u%data_3d); call hip_upload(hip, ICAR_F_V, this%v%data_3d); call hip_upload(hip, ICAR_F_W, this%w%data_3d)
    call hip_upload(hip, ICAR_F_PRESSURE, this%pressure%data_3d); call hip_upload(hip, ICAR_F_EXNER, this%exner%data_3d)
    call hip_upload(hip, ICAR_F_DENSITY,  this%density%data_3d)
    call hip_download(hip, ICAR_F_WATER_VAPOR, this%water_vapor%data_3d)
    call hip_download_2dd(hip, ICAR_F_PRECIPITATION, this%accumulated_precipitation%data_2dd)
    call hip_dqdt_upload(hip, ICAR_F_U, this%u%meta_data%dqdt_3d)
  end subroutine

  !> section 3: the fixed dispatch order of adv_mpdata.f90:512-522 from options%vars_to_advect(kVARS%...)
  subroutine glue_vars(options, vars, nv)
    type(options_t), intent(in) :: options
    integer(c_int), intent(out) :: vars(ICAR_N_ADVECTABLE)
    integer, intent(out) :: nv
    integer :: i, kv(0:ICAR_N_ADVECTABLE-1)
    kv = [kVARS%water_vapor, kVARS%cloud_water, kVARS%rain_in_air, kVARS%snow_in_air, kVARS%potential_temperature, kVARS%cloud_ice, &
          kVARS%graupel_in_air, kVARS%ice_number_concentration, kVARS%rain_number_concentration, kVARS%snow_number_concentration, &
          kVARS%graupel_number_concentration]                                                          ! kvars_of_field
    nv = 0
    do i = 0, ICAR_N_ADVECTABLE-1
        if (options%vars_to_advect(kv(i)) > 0) then; nv = nv+1; vars(nv) = i; endif                   ! [INTEGRATION section 3]
    enddo
  end subroutine

  !> section 3a: the whole of step() as one call (main/time_step.f90:440-551)
  subroutine glue_step(domain, hip, end_time, options)
    type(domain_t), intent(inout) :: domain
    type(hip_ctx_t), intent(in) :: hip
    type(Time_type), intent(in) :: end_time
    type(options_t), intent(in) :: options
    type(hip_step_config_t) :: cfg
    integer(c_int) :: vars(ICAR_N_ADVECTABLE), forced_ids(4)
    logical :: forced_is_boundary_only(4)
    integer :: nv, nf, nsteps
    type(time_delta_t) :: elapsed
    real(c_double) :: t0
    call glue_vars(options, vars, nv)
    nf = 4; forced_ids = [ICAR_F_U, ICAR_F_V, ICAR_F_W, ICAR_F_PRESSURE]; forced_is_boundary_only = .false.
    cfg%advection = options%physics%advection;            cfg%microphysics = options%physics%microphysics
    cfg%mpdata_order = options%adv_options%mpdata_order;  cfg%flux_corrected_transport = merge(1, 0, options%adv_options%flux_corrected_transport)
    cfg%advect_density = merge(1, 0, options%parameters%advect_density)
    cfg%cfl_strictness = options%parameters%cfl_strictness; cfg%cfl_reduction_factor = options%parameters%cfl_reduction_factor
    cfg%dx = domain%dx;  cfg%mp_update_interval = options%mp_options%update_interval;  cfg%top_mp_level = options%mp_options%top_mp_level
    cfg%halo_size = domain%grid%halo_size
    cfg%its = domain%grid%its; cfg%ite = domain%grid%ite; cfg%jts = domain%grid%jts; cfg%jte = domain%grid%jte
    cfg%kts = domain%grid%kts; cfg%kte = domain%grid%kte; cfg%ids = domain%grid%ids; cfg%ide = domain%grid%ide
    cfg%jds = domain%grid%jds; cfg%jde = domain%grid%jde; cfg%kds = domain%grid%kds; cfg%kde = domain%grid%kde
    cfg%west_boundary  = merge(1, 0, domain%grid%ximg == 1); cfg%east_boundary  = merge(1, 0, domain%grid%ximg == domain%grid%ximages)
    cfg%south_boundary = merge(1, 0, domain%grid%yimg == 1); cfg%north_boundary = merge(1, 0, domain%grid%yimg == domain%grid%yimages)
    cfg%n_advect = nv;   cfg%advect_fields(1:nv)   = vars(1:nv)
    cfg%n_exchange = nv; cfg%exchange_fields(1:nv) = vars(1:nv)
    cfg%n_forced = nf;   cfg%forced_fields(1:nf) = forced_ids; cfg%force_boundaries(1:nf) = merge(1, 0, forced_is_boundary_only)
    call hip_step_configure(hip, cfg, options%parameters%dz_levels)                                    ! [INTEGRATION section 3a]
    t0 = real(domain%model_time%seconds(), c_double)                        ! Time_type%seconds() is real128 (utilities/time_h.f90:151)
    call hip_set_model_time(hip, t0)
    nsteps = hip_step(hip, real(end_time%seconds(), c_double))
    call elapsed%set(seconds = hip_model_time(hip) - t0)                    ! time_delta_t%set_time_delta_d (real64 seconds)
    domain%model_time = domain%model_time + elapsed                         ! time_step.f90:547, once for the interval
  end subroutine

  !> section 3: the call sites in step(), one level down (dt is the time_delta_t of step(), its..kde the tile bounds of mp())
  subroutine glue_substep(domain, hip, options, dt, its,ite, jts,jte, kts,kte, ids,ide, jds,jde, kds,kde)
    type(domain_t), intent(inout) :: domain
    type(hip_ctx_t), intent(in) :: hip
    type(options_t), intent(in) :: options
    type(time_delta_t), intent(in) :: dt
    integer, intent(in) :: its,ite, jts,jte, kts,kte, ids,ide, jds,jde, kds,kde
    integer(c_int) :: vars(ICAR_N_ADVECTABLE), forced_ids(4)
    logical :: forced_is_boundary_only(4)
    integer :: nv
    double precision :: seconds
    forced_ids = [ICAR_F_U, ICAR_F_V, ICAR_F_W, ICAR_F_PRESSURE]; forced_is_boundary_only = .false.
    seconds = options%parameters%cfl_reduction_factor / hip_max_courant(hip, domain%dx, options%parameters%dz_levels)
    call hip_co_min(hip, seconds)
    call hip_diagnostic_update(hip)
    if (options%physics%microphysics == kMP_THOMPSON) then
        call hip_thompson(hip, real(dt%seconds()), its,ite, jts,jte, kts,kte, ids,ide, jds,jde, kds,kde)
    elseif (options%physics%microphysics == kMP_SB04) then
        call hip_mp_simple(hip, real(dt%seconds()), its,ite, jts,jte, kts,kte)
    elseif (options%physics%microphysics == kMP_WSM3) then
        call hip_wsm3(hip, real(dt%seconds()), its,ite, jts,jte, kts,kte)
    elseif (options%physics%microphysics == kMP_WSM6) then
        call hip_wsm6(hip, real(dt%seconds()), its,ite, jts,jte, kts,kte)
    endif
    call glue_vars(options, vars, nv)
    call hip_advect(hip, options%physics%advection, options%adv_options%mpdata_order, &
                    options%adv_options%flux_corrected_transport, options%parameters%advect_density, real(dt%seconds()), domain%dx, vars(1:nv))
    call hip_apply_forcing(hip, dt%seconds(), forced_ids, forced_is_boundary_only, &
                           west=domain%grid%ximg==1, east=domain%grid%ximg==domain%grid%ximages, &
                           south=domain%grid%yimg==1, north=domain%grid%yimg==domain%grid%yimages)
    call hip_enforce_limits(hip, vars(1:nv))
  end subroutine

  !> physics/mp_driver.f90:80  mp_init -> thompson_init(options%mp_options)
  subroutine glue_mp_init(hip, options)
    type(hip_ctx_t), intent(in) :: hip
    type(options_t), intent(in) :: options
    associate (mpo => options%mp_options)
    call hip_thompson_init(hip, [mpo%Nt_c, mpo%TNO, mpo%am_s, mpo%rho_g, mpo%av_s, mpo%bv_s, mpo%fv_s, mpo%av_g, &
         mpo%bv_g, mpo%av_i, mpo%Ef_si, mpo%Ef_rs, mpo%Ef_rg, mpo%Ef_ri, mpo%C_cubes, mpo%C_sqrd, mpo%mu_r, mpo%t_adjust], &
         mpo%Ef_rw_l, mpo%Ef_sw_l)                                                                      ! [INTEGRATION section 3]
    end associate
    if (options%physics%microphysics == kMP_WSM3) call hip_wsm3_init(hip)
    if (options%physics%microphysics == kMP_WSM6) call hip_wsm6_init(hip)
  end subroutine

  !> section 3b: linear-theory winds and update_winds as one call
  subroutine glue_winds(domain, hip, options, updt)
    type(domain_t), intent(inout) :: domain
    type(hip_ctx_t), intent(in) :: hip
    type(options_t), intent(in) :: options
    logical, intent(in) :: updt
    type(hip_lt_options_t) :: lt
    real, allocatable :: layer_height(:)
    integer :: ims, jms, nz, it, halo
    ims = domain%grid%ims; jms = domain%grid%jms; nz = domain%grid%nz; halo = domain%grid%halo_size
    associate (opt => options%lt_options)
    lt = hip_lt_options_t(opt%buffer, opt%stability_window_size, opt%vert_smooth, merge(1,0,opt%variable_N), merge(1,0,opt%smooth_nsq), &
                          opt%max_stability, opt%min_stability, opt%N_squared, opt%linear_contribution, opt%linear_update_fraction, &
                          opt%dirmax, opt%dirmin, opt%spdmax, opt%spdmin, opt%nsqmax, opt%nsqmin, &
                          opt%n_dir_values, opt%n_nsq_values, opt%n_spd_values, opt%minimum_layer_size)   ! [INTEGRATION section 3b]
    end associate
    call hip_setup_linwinds(hip, lt, domain%global_terrain, domain%grid%ids, domain%grid%jds, domain%dx)
    layer_height = domain%z%data_3d(ims,:,jms) - domain%terrain%data_2d(ims,jms)
    call hip_linwinds_build_lut(hip, layer_height - options%parameters%dz_levels(:nz)/2, layer_height + options%parameters%dz_levels(:nz)/2)
    call hip_spatial_winds(hip, updt)
    call hip_upload(hip, ICAR_F_Z, domain%z%data_3d)
    call hip_upload_2dd(hip, ICAR_F_SINTHETA, domain%sintheta); call hip_upload_2dd(hip, ICAR_F_COSTHETA, domain%costheta)
    call hip_update_winds(hip, options%physics%windtype, options%parameters%wind_iterations, domain%dx, domain%grid%halo_size)
    ! the loop of iterative_winds (wind.f90:371-498) in the host's own hands
    call hip_exchange_uv(hip, halo, updt)
    if (updt) then
        call hip_balance_uvw_update(hip, domain%dx)
    else
        call hip_balance_uvw(hip, domain%dx)
    endif
    call hip_iterative_winds_correct_w(hip, updt)
    do it = 0, options%parameters%wind_iterations
        call hip_iterative_winds_sweep(hip, domain%dx, 1, updt)
        call hip_exchange_uv(hip, halo, updt)
    enddo
  end subroutine

  !> section 4: the communicator (objects/domain_obj.f90, end of domain%init; replaces exchangeable%set_neighbors) and the exchanges
  subroutine glue_comm(this, hip, vars, nv)
    type(domain_t), intent(in) :: this
    type(hip_ctx_t), intent(in) :: hip
    integer(c_int), intent(in) :: vars(ICAR_N_ADVECTABLE)
    integer, intent(in) :: nv
    character(kind=c_char) :: uid(128)
    integer(c_int) :: nb(4)
    double precision :: seconds
    nb = ICAR_NEIGHBOR_NONE
    if (.not. this%north_boundary) nb(1) = this_image() + this%grid%ximages - 1
    if (.not. this%south_boundary) nb(2) = this_image() - this%grid%ximages - 1
    if (.not. this%east_boundary)  nb(3) = this_image()
    if (.not. this%west_boundary)  nb(4) = this_image() - 2
    if (this_image() == 1) call hip_comm_unique_id(uid)
    call co_broadcast(uid, 1)
    call hip_comm_init(hip, num_images(), this_image() - 1, uid, nb)
    call hip_halo_send(hip, this%grid%halo_size, vars(1:nv))
    call hip_halo_retrieve(hip, this%grid%halo_size, vars(1:nv))
    seconds = 60.0d0
    call hip_co_min(hip, seconds)
    if (hip_comm_ranks(hip) /= num_images()) error stop "communicator does not span the images"
    if (hip_halo_selfcheck(hip, this%grid%halo_size) /= 0) error stop "halo exchange self-test failed"
  end subroutine
end module integration_glue
