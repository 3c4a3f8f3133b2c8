!> oracle/ref_shim.f90 -- TEST INFRASTRUCTURE, NOT PRODUCT.
!!
!! bind(C) entry points around the *unmodified reference kernels* compiled from
!! /root/reference/src by oracle/build_ref.sh into oracle/_ref/libicar_ref.so.
!! This file is our own code: it only fills a reference `domain_t`/`options_t`
!! from plain C arrays and calls the reference's public procedures
!!   mpdata(domain,options,dt)          src/physics/adv_mpdata.f90:463
!!   upwind(domain,options,dt)          src/physics/advect.f90:380
!!   mp_simple_driver(...)              src/physics/mp_simple.f90:595
!!   thompson_init(mp_options)          src/physics/mp_thompson.f90:342
!!   mp_gt_driver(...)                  src/physics/mp_thompson.f90:772
!! All arrays are Fortran order (i,k,j), REAL(4), lower bound 1.
!!
!! NOTE (module SAVE state in the reference): adv_mpdata/adv_upwind keep private
!! allocatable U_m,V_m,W_m sized on first call, so ONE grid size per process.
module icar_ref_shim
  use iso_c_binding
  use icar_constants
  use options_interface, only: options_t
  use options_types,     only: mp_options_type
  use domain_interface,  only: domain_t
  use grid_interface,    only: grid_t
  use mod_atm_utilities, only: exner_function, calc_direction, calc_speed, calc_u, calc_v, calc_stability, &
                               compute_ivt, compute_iq, sat_mr
  use array_utilities,   only: smooth_array, linear_space, calc_weight
  use module_mp_wsm3,    only: wsm3, wsm3init, w3_qc0 => qc0, w3_qck1 => qck1, w3_pidnc => pidnc, w3_bvtr1 => bvtr1, &
       w3_bvtr2 => bvtr2, w3_bvtr3 => bvtr3, w3_bvtr4 => bvtr4, w3_g1pbr => g1pbr, w3_g3pbr => g3pbr, w3_g4pbr => g4pbr, &
       w3_g5pbro2 => g5pbro2, w3_pvtr => pvtr, w3_eacrr => eacrr, w3_pacrr => pacrr, w3_precr1 => precr1, w3_precr2 => precr2, &
       w3_xmmax => xmmax, w3_roqimax => roqimax, w3_bvts1 => bvts1, w3_bvts2 => bvts2, w3_bvts3 => bvts3, w3_bvts4 => bvts4, &
       w3_g1pbs => g1pbs, w3_g3pbs => g3pbs, w3_g4pbs => g4pbs, w3_g5pbso2 => g5pbso2, w3_pvts => pvts, w3_pacrs => pacrs, &
       w3_precs1 => precs1, w3_precs2 => precs2, w3_pidn0r => pidn0r, w3_pidn0s => pidn0s, w3_xlv1 => xlv1, w3_pi => pi, &
       w3_rslopermax => rslopermax, w3_rslopesmax => rslopesmax, w3_rsloperbmax => rsloperbmax, w3_rslopesbmax => rslopesbmax, &
       w3_rsloper2max => rsloper2max, w3_rslopes2max => rslopes2max, w3_rsloper3max => rsloper3max, w3_rslopes3max => rslopes3max
  use module_mp_wsm6,    only: wsm6, wsm6init, &
       w6_qc0 => qc0, w6_qck1 => qck1, w6_bvtr1 => bvtr1, w6_bvtr2 => bvtr2, w6_bvtr3 => bvtr3, w6_bvtr4 => bvtr4, &
       w6_g1pbr => g1pbr, w6_g3pbr => g3pbr, w6_g4pbr => g4pbr, w6_g5pbro2 => g5pbro2, w6_pvtr => pvtr, w6_eacrr => eacrr, &
       w6_pacrr => pacrr, w6_bvtr6 => bvtr6, w6_g6pbr => g6pbr, w6_precr1 => precr1, w6_precr2 => precr2, &
       w6_roqimax => roqimax, w6_bvts1 => bvts1, w6_bvts2 => bvts2, w6_bvts3 => bvts3, w6_bvts4 => bvts4, w6_g1pbs => g1pbs, &
       w6_g3pbs => g3pbs, w6_g4pbs => g4pbs, w6_g5pbso2 => g5pbso2, w6_pvts => pvts, w6_pacrs => pacrs, w6_precs1 => precs1, &
       w6_precs2 => precs2, w6_pidn0r => pidn0r, w6_pidn0s => pidn0s, w6_xlv1 => xlv1, w6_pacrc => pacrc, w6_pi => pi, &
       w6_bvtg1 => bvtg1, w6_bvtg2 => bvtg2, w6_bvtg3 => bvtg3, w6_bvtg4 => bvtg4, w6_g1pbg => g1pbg, w6_g3pbg => g3pbg, &
       w6_g4pbg => g4pbg, w6_g5pbgo2 => g5pbgo2, w6_pvtg => pvtg, w6_pacrg => pacrg, w6_precg1 => precg1, &
       w6_precg2 => precg2, w6_pidn0g => pidn0g, w6_rslopermax => rslopermax, w6_rslopesmax => rslopesmax, &
       w6_rslopegmax => rslopegmax, w6_rsloperbmax => rsloperbmax, w6_rslopesbmax => rslopesbmax, &
       w6_rslopegbmax => rslopegbmax, w6_rsloper2max => rsloper2max, w6_rslopes2max => rslopes2max, &
       w6_rslopeg2max => rslopeg2max, w6_rsloper3max => rsloper3max, w6_rslopes3max => rslopes3max, &
       w6_rslopeg3max => rslopeg3max
  use mod_wrf_constants, only: wc_cpv => cpv, wc_cliq => cliq, wc_cice => cice, wc_psat => psat, wc_XLS => XLS, wc_XLV => XLV, &
       wc_XLF => XLF, wc_rhoair0 => rhoair0, wc_rhowater => rhowater, wc_rhosnow => rhosnow, wc_epsilon => epsilon
  use prif,              only: stub_num_images
  use adv_mpdata,        only: mpdata
  use adv_upwind,        only: upwind
  use module_mp_simple,  only: mp_simple_driver
  use module_mp_thompson,only: thompson_init, mp_gt_driver, &
       tcg_racg, tmr_racg, tcr_gacr, tmg_gacr, tnr_racg, tnr_gacr, tcs_racs1, tmr_racs1, tcs_racs2, tmr_racs2, &
       tcr_sacr1, tms_sacr1, tcr_sacr2, tms_sacr2, tnr_racs1, tnr_racs2, tnr_sacr1, tnr_sacr2, tpi_qcfz, tni_qcfz, &
       tpi_qrfz, tpg_qrfz, tni_qrfz, tnr_qrfz, tps_iaus, tni_iaus, tpi_ide, t_Efrw, t_Efsw, &
       t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi, t1_qr_ev, t2_qr_ev, t1_qs_sd, t2_qs_sd, &
       t1_qg_sd, t2_qg_sd, t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me, Dc, Di, Dr, Ds, Dg, dtc, dti, dtr, dts, dtg
  implicit none
  type(options_t), save :: options
  type(domain_t), allocatable, save :: domain
  integer, save :: cur_nx = -1, cur_nz = -1, cur_ny = -1
contains

  !> scheme: 1 = upwind (kADV_UPWIND), 2 = mpdata (kADV_MPDATA)
  !! q is (nx,nz,ny,nvars): each variable is advected by pointing
  !! domain%water_vapor%data_3d at it (every advected scalar goes through the same advect3d).
  subroutine ref_advect(scheme, nx, nz, ny, nvars, q, u, v, w, rho, jaco, jaco_u, jaco_v, jaco_w, &
                        dz3d, dz_levels, dx, dt, advect_density, mpdata_order, fct, nsteps) bind(C, name="ref_advect")
    integer(c_int), value :: scheme, nx, nz, ny, nvars, advect_density, mpdata_order, fct, nsteps
    real(c_float), value :: dx, dt
    real(c_float), target, intent(inout) :: q(nx,nz,ny,nvars)
    real(c_float), target, intent(in) :: u(nx+1,nz,ny), v(nx,nz,ny+1), w(nx,nz,ny), rho(nx,nz,ny)
    real(c_float), intent(in) :: jaco(nx,nz,ny), jaco_u(nx+1,nz,ny), jaco_v(nx,nz,ny+1), jaco_w(nx,nz,ny)
    real(c_float), intent(in) :: dz3d(nx,nz,ny), dz_levels(nz)
    integer :: n, s

    if (.not.allocated(domain)) allocate(domain)
    if (cur_nx /= -1 .and. (cur_nx/=nx .or. cur_nz/=nz .or. cur_ny/=ny)) then
       print *, "ref_advect: reference module state is sized for one grid per process"
       error stop
    endif
    cur_nx = nx; cur_nz = nz; cur_ny = ny
    domain%grid%ims=1; domain%grid%ime=nx; domain%grid%jms=1; domain%grid%jme=ny; domain%grid%kms=1; domain%grid%kme=nz
    domain%ims=1; domain%ime=nx; domain%jms=1; domain%jme=ny; domain%kms=1; domain%kme=nz
    domain%dx = dx
    domain%u%data_3d => u
    domain%v%data_3d => v
    domain%w%data_3d => w
    domain%density%data_3d => rho
    if (.not.allocated(domain%jacobian)) then
       allocate(domain%jacobian(nx,nz,ny), domain%jacobian_u(nx+1,nz,ny), domain%jacobian_v(nx,nz,ny+1), &
                domain%jacobian_w(nx,nz,ny), domain%advection_dz(nx,nz,ny))
    endif
    domain%jacobian = jaco; domain%jacobian_u = jaco_u; domain%jacobian_v = jaco_v; domain%jacobian_w = jaco_w
    domain%advection_dz = dz3d
    options%parameters%dz_levels(1:nz) = dz_levels
    options%parameters%advect_density = (advect_density /= 0)
    options%parameters%debug = .false.
    options%vars_to_advect = 0
    options%vars_to_advect(kVARS%water_vapor) = 1
    options%adv_options%mpdata_order = mpdata_order
    options%adv_options%flux_corrected_transport = (fct /= 0)
    do s = 1, nsteps
      do n = 1, nvars
        domain%water_vapor%data_3d => q(:,:,:,n)
        if (scheme == 1) then
          call upwind(domain, options, dt)
        else
          call mpdata(domain, options, dt)
        endif
      enddo
    enddo
  end subroutine

  subroutine ref_mp_simple(nx, nz, ny, pressure, th, pii, rho, qv, qc, qr, qs, rain, snow, dt, dz, &
                           its, ite, jts, jte, kts, kte) bind(C, name="ref_mp_simple")
    integer(c_int), value :: nx, nz, ny, its, ite, jts, jte, kts, kte
    real(c_float), value :: dt
    real(c_float), intent(inout) :: pressure(nx,nz,ny), th(nx,nz,ny), pii(nx,nz,ny), rho(nx,nz,ny)
    real(c_float), intent(inout) :: qv(nx,nz,ny), qc(nx,nz,ny), qr(nx,nz,ny), qs(nx,nz,ny)
    real(c_float), intent(inout) :: rain(nx,ny), snow(nx,ny)
    real(c_float), intent(in) :: dz(nx,nz,ny)
    call mp_simple_driver(pressure, th, pii, rho, qv, qc, qr, qs, rain, snow, dt, dz, &
                          1, nx, 1, ny, 1, nz, its, ite, jts, jte, kts, kte)
  end subroutine

  !> params(18): Nt_c,TNO,am_s,rho_g,av_s,bv_s,fv_s,av_g,bv_g,av_i,Ef_si,Ef_rs,Ef_rg,Ef_ri,C_cubes,C_sqrd,mu_r,t_adjust
  !! flags(2): Ef_rw_l, Ef_sw_l.  Writes/reads the reference's *.dat table caches in the CWD.
  subroutine ref_thompson_init(params, flags) bind(C, name="ref_thompson_init")
    real(c_float), intent(in) :: params(18)
    integer(c_int), intent(in) :: flags(2)
    type(mp_options_type) :: mpo
    mpo%Nt_c=params(1); mpo%TNO=params(2); mpo%am_s=params(3); mpo%rho_g=params(4)
    mpo%av_s=params(5); mpo%bv_s=params(6); mpo%fv_s=params(7); mpo%av_g=params(8); mpo%bv_g=params(9)
    mpo%av_i=params(10); mpo%Ef_si=params(11); mpo%Ef_rs=params(12); mpo%Ef_rg=params(13); mpo%Ef_ri=params(14)
    mpo%C_cubes=params(15); mpo%C_sqrd=params(16); mpo%mu_r=params(17); mpo%t_adjust=params(18)
    mpo%Ef_rw_l = (flags(1)/=0); mpo%Ef_sw_l = (flags(2)/=0)
    mpo%update_interval = 0; mpo%top_mp_level = 0; mpo%local_precip_fraction = 1
    call thompson_init(mpo)
  end subroutine

  subroutine ref_thompson(nx, nz, ny, qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, dt, &
                          rainnc, rainncv, snownc, graupelnc, sr, &
                          ids, ide, jds, jde, kds, kde, its, ite, jts, jte, kts, kte) bind(C, name="ref_thompson")
    integer(c_int), value :: nx, nz, ny, ids, ide, jds, jde, kds, kde, its, ite, jts, jte, kts, kte
    real(c_float), value :: dt
    real(c_float), intent(inout), dimension(nx,nz,ny) :: qv, qc, qr, qi, qs, qg, ni, nr, th
    real(c_float), intent(in), dimension(nx,nz,ny) :: pii, p, dz
    real(c_float), intent(inout), dimension(nx,ny) :: rainnc, rainncv, snownc, graupelnc, sr
    call mp_gt_driver(qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, dt, 1, rainnc, rainncv, &
                      SNOWNC=snownc, GRAUPELNC=graupelnc, SR=sr, &
                      ids=ids, ide=ide, jds=jds, jde=jde, kds=kds, kde=kde, &
                      ims=1, ime=nx, jms=1, jme=ny, kms=1, kme=nz, &
                      its=its, ite=ite, jts=jts, jte=jte, kts=kts, kte=kte)
  end subroutine

  !> Copy one of the reference's (public) Thompson lookup tables / constants into out(n). Returns the
  !! number of elements, or -1 for an unknown id.  ids follow oracle/ref.py:THOMPSON_TABLES.
  integer(c_int) function ref_thompson_table(id, n, out) bind(C, name="ref_thompson_table")
    integer(c_int), value :: id, n
    real(c_double), intent(out) :: out(n)
    integer :: m
    m = -1
    select case (id)
    case (1);  m = size(tcg_racg);  if (m<=n) out(1:m) = reshape(tcg_racg, [m])
    case (2);  m = size(tmr_racg);  if (m<=n) out(1:m) = reshape(tmr_racg, [m])
    case (3);  m = size(tcr_gacr);  if (m<=n) out(1:m) = reshape(tcr_gacr, [m])
    case (4);  m = size(tmg_gacr);  if (m<=n) out(1:m) = reshape(tmg_gacr, [m])
    case (5);  m = size(tnr_racg);  if (m<=n) out(1:m) = reshape(tnr_racg, [m])
    case (6);  m = size(tnr_gacr);  if (m<=n) out(1:m) = reshape(tnr_gacr, [m])
    case (7);  m = size(tcs_racs1); if (m<=n) out(1:m) = reshape(tcs_racs1, [m])
    case (8);  m = size(tmr_racs1); if (m<=n) out(1:m) = reshape(tmr_racs1, [m])
    case (9);  m = size(tcs_racs2); if (m<=n) out(1:m) = reshape(tcs_racs2, [m])
    case (10); m = size(tmr_racs2); if (m<=n) out(1:m) = reshape(tmr_racs2, [m])
    case (11); m = size(tcr_sacr1); if (m<=n) out(1:m) = reshape(tcr_sacr1, [m])
    case (12); m = size(tms_sacr1); if (m<=n) out(1:m) = reshape(tms_sacr1, [m])
    case (13); m = size(tcr_sacr2); if (m<=n) out(1:m) = reshape(tcr_sacr2, [m])
    case (14); m = size(tms_sacr2); if (m<=n) out(1:m) = reshape(tms_sacr2, [m])
    case (15); m = size(tnr_racs1); if (m<=n) out(1:m) = reshape(tnr_racs1, [m])
    case (16); m = size(tnr_racs2); if (m<=n) out(1:m) = reshape(tnr_racs2, [m])
    case (17); m = size(tnr_sacr1); if (m<=n) out(1:m) = reshape(tnr_sacr1, [m])
    case (18); m = size(tnr_sacr2); if (m<=n) out(1:m) = reshape(tnr_sacr2, [m])
    case (19); m = size(tpi_qcfz);  if (m<=n) out(1:m) = reshape(tpi_qcfz, [m])
    case (20); m = size(tni_qcfz);  if (m<=n) out(1:m) = reshape(tni_qcfz, [m])
    case (21); m = size(tpi_qrfz);  if (m<=n) out(1:m) = reshape(tpi_qrfz, [m])
    case (22); m = size(tpg_qrfz);  if (m<=n) out(1:m) = reshape(tpg_qrfz, [m])
    case (23); m = size(tni_qrfz);  if (m<=n) out(1:m) = reshape(tni_qrfz, [m])
    case (24); m = size(tnr_qrfz);  if (m<=n) out(1:m) = reshape(tnr_qrfz, [m])
    case (25); m = size(tps_iaus);  if (m<=n) out(1:m) = reshape(tps_iaus, [m])
    case (26); m = size(tni_iaus);  if (m<=n) out(1:m) = reshape(tni_iaus, [m])
    case (27); m = size(tpi_ide);   if (m<=n) out(1:m) = reshape(tpi_ide, [m])
    case (28); m = size(t_Efrw);    if (m<=n) out(1:m) = reshape(t_Efrw, [m])
    case (29); m = size(t_Efsw);    if (m<=n) out(1:m) = reshape(t_Efsw, [m])
    case (30)
      m = 16
      if (m<=n) out(1:16) = [t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi, t1_qr_ev, t2_qr_ev, t1_qs_sd, t2_qs_sd, &
                             t1_qg_sd, t2_qg_sd, t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me]
    case (31); m = 1000
      if (m<=n) then
        out(1:100)=Dc; out(101:200)=dtc; out(201:300)=Di; out(301:400)=dti; out(401:500)=Dr; out(501:600)=dtr
        out(601:700)=Ds; out(701:800)=dts; out(801:900)=Dg; out(901:1000)=dtg
      endif
    end select
    ref_thompson_table = m
  end function

  !> grid_t%set_grid_dimensions(nx, ny, nz, nx_extra, ny_extra, for_image=image) of an nimages-image run
  !! (src/objects/grid_obj.f90:140-255).  out(1:33) = yimg,ximg,yimages,ximages, ims,ime,jms,jme,kms,kme,
  !! ns_halo_nx,ew_halo_ny,halo_nz,halo_size, nx_global,ny_global, nx,ny,nz, ids,ide,jds,jde,kds,kde,
  !! its,ite,jts,jte,kts,kte, is2d,is3d
  subroutine ref_grid(nx, ny, nz, nimages, image, nx_extra, ny_extra, out) bind(C, name="ref_grid")
    integer(c_int), value :: nx, ny, nz, nimages, image, nx_extra, ny_extra
    integer(c_int), intent(out) :: out(33)
    type(grid_t) :: g
    stub_num_images = nimages
    call g%set_grid_dimensions(nx, ny, nz, nx_extra=nx_extra, ny_extra=ny_extra, for_image=image)
    stub_num_images = 1
    out = [g%yimg, g%ximg, g%yimages, g%ximages, g%ims, g%ime, g%jms, g%jme, g%kms, g%kme, &
           g%ns_halo_nx, g%ew_halo_ny, g%halo_nz, g%halo_size, g%nx_global, g%ny_global, g%nx, g%ny, g%nz, &
           g%ids, g%ide, g%jds, g%jde, g%kds, g%kde, g%its, g%ite, g%jts, g%jte, g%kts, g%kte, &
           merge(1, 0, g%is2d), merge(1, 0, g%is3d)]
  end subroutine

  !> helpers of src/utilities/atm_utilities.f90 (exner_function :682, calc_direction :334, calc_speed :361, calc_u :373,
  !! calc_v :385, calc_stability :448, compute_ivt :35, compute_iq :73) and src/utilities/array_utilities.f90
  !! (linear_space :215, calc_weight :263, smooth_array_3d :308) that rows T3 and W2 call -- element-wise over n values
  subroutine ref_exner(n, p, out) bind(C, name="ref_exner")
    integer(c_int), value :: n
    real(c_float), intent(in) :: p(n)
    real(c_float), intent(out) :: out(n)
    out = exner_function(p)
  end subroutine

  subroutine ref_wind_polar(n, u, v, direction, speed, u_back, v_back) bind(C, name="ref_wind_polar")
    integer(c_int), value :: n
    real(c_float), intent(in) :: u(n), v(n)
    real(c_float), intent(out) :: direction(n), speed(n), u_back(n), v_back(n)
    integer :: i
    do i = 1, n
       direction(i) = calc_direction(u(i), v(i))
    end do
    speed = calc_speed(u, v)
    u_back = calc_u(direction, speed)
    v_back = calc_v(direction, speed)
  end subroutine

  subroutine ref_calc_stability(n, th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc, out) bind(C, name="ref_calc_stability")
    integer(c_int), value :: n
    real(c_float), intent(in), dimension(n) :: th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc
    real(c_float), intent(out) :: out(n)
    integer :: i
    do i = 1, n
       out(i) = calc_stability(th_top(i), th_bot(i), pii_top(i), pii_bot(i), z_top(i), z_bot(i), qv_top(i), qv_bot(i), qc(i))
    end do
  end subroutine

  subroutine ref_compute_ivt(nx, nz, ny, qv, u, v, p_i, out) bind(C, name="ref_compute_ivt")
    integer(c_int), value :: nx, nz, ny
    real(c_float), intent(in), dimension(nx,nz,ny) :: qv, u, v, p_i
    real(c_float), intent(out) :: out(nx,ny)
    call compute_ivt(out, qv, u, v, p_i)
  end subroutine

  subroutine ref_compute_iq(nx, nz, ny, q, p_i, out) bind(C, name="ref_compute_iq")
    integer(c_int), value :: nx, nz, ny
    real(c_float), intent(in), dimension(nx,nz,ny) :: q, p_i
    real(c_float), intent(out) :: out(nx,ny)
    call compute_iq(out, q, p_i)
  end subroutine

  subroutine ref_linear_space(n, vmin, vmax, out) bind(C, name="ref_linear_space")
    integer(c_int), value :: n
    real(c_float), value :: vmin, vmax
    real(c_float), intent(out) :: out(n)
    real, allocatable :: a(:)
    call linear_space(a, vmin, vmax, n)
    out = a
  end subroutine

  !> calc_weight for m (bestpos, match) pairs on one axis; nextpos comes back 1-based like the reference's
  subroutine ref_calc_weight(n, axis, m, bestpos, match, nextpos, weight) bind(C, name="ref_calc_weight")
    integer(c_int), value :: n, m
    real(c_float), intent(in) :: axis(n), match(m)
    integer(c_int), intent(in) :: bestpos(m)
    integer(c_int), intent(out) :: nextpos(m)
    real(c_float), intent(out) :: weight(m)
    integer :: i
    do i = 1, m
       nextpos(i) = -1
       weight(i) = calc_weight(axis, bestpos(i), nextpos(i), match(i))
    end do
  end subroutine

  subroutine ref_smooth_array_3d(nx, nz, ny, wind, windowsize, ydim) bind(C, name="ref_smooth_array_3d")
    integer(c_int), value :: nx, nz, ny, windowsize, ydim
    real(c_float), intent(inout) :: wind(nx,nz,ny)
    call smooth_array(wind, windowsize, ydim)
  end subroutine

  !> wsm3init (mp_wsm3.f90:951-1006) as mp_driver.f90:105 calls it; out(1:44) = the module constants it derives, in the order
  !! of the use statement above; args(1:18) = what mp_driver.f90:554-585 passes to wsm3:
  !! delt(unused here), g, cpd, cpv, rd, rv, t0c, ep1, ep2, qmin, XLS, XLV0, XLF0, den0, denr, cliq, cice, psat
  subroutine ref_wsm3_init(out, args) bind(C, name="ref_wsm3_init")
    real(c_float), intent(out) :: out(44), args(18)
    call wsm3init(wc_rhoair0, wc_rhowater, wc_rhosnow, wc_cliq, wc_cpv, allowed_to_read=.true.)
    out = [w3_qc0, w3_qck1, w3_pidnc, w3_bvtr1, w3_bvtr2, w3_bvtr3, w3_bvtr4, w3_g1pbr, w3_g3pbr, w3_g4pbr, w3_g5pbro2, w3_pvtr, &
           w3_eacrr, w3_pacrr, w3_precr1, w3_precr2, w3_xmmax, w3_roqimax, w3_bvts1, w3_bvts2, w3_bvts3, w3_bvts4, w3_g1pbs, &
           w3_g3pbs, w3_g4pbs, w3_g5pbso2, w3_pvts, w3_pacrs, w3_precs1, w3_precs2, w3_pidn0r, w3_pidn0s, w3_xlv1, w3_pi, &
           w3_rslopermax, w3_rslopesmax, w3_rsloperbmax, w3_rslopesbmax, w3_rsloper2max, w3_rslopes2max, w3_rsloper3max, &
           w3_rslopes3max, 0.0, 0.0]
    args = [0.0, gravity, cp, wc_cpv, Rd, Rw, 273.15, EP1, EP2, wc_epsilon, wc_XLS, wc_XLV, wc_XLF, wc_rhoair0, wc_rhowater, &
            wc_cliq, wc_cice, wc_psat]
  end subroutine

  !> wsm3 (mp_wsm3.f90:74) on a tile exactly as mp_driver.f90:554-585 calls it (has_req* = 0).  Arrays (nx,nz,ny) / (nx,ny).
  subroutine ref_wsm3(nx, nz, ny, th, q, qci, qrs, w, den, pii, p, delz, delt, rain, rainncv, snow, snowncv, sr, &
                      its, ite, jts, jte, kts, kte) bind(C, name="ref_wsm3")
    integer(c_int), value :: nx, nz, ny, its, ite, jts, jte, kts, kte
    real(c_float), value :: delt
    real(c_float), intent(inout), dimension(nx,nz,ny) :: th, q, qci, qrs
    real(c_float), intent(in), dimension(nx,nz,ny) :: w, den, pii, p, delz
    real(c_float), intent(inout), dimension(nx,ny) :: rain, rainncv, snow, snowncv, sr
    call wsm3(th=th, q=q, qci=qci, qrs=qrs, w=w, den=den, pii=pii, p=p, delz=delz, delt=delt, g=gravity, cpd=cp, cpv=wc_cpv, &
              rd=Rd, rv=Rw, t0c=273.15, ep1=EP1, ep2=EP2, qmin=wc_epsilon, XLS=wc_XLS, XLV0=wc_XLV, XLF0=wc_XLF, &
              den0=wc_rhoair0, denr=wc_rhowater, cliq=wc_cliq, cice=wc_cice, psat=wc_psat, rain=rain, rainncv=rainncv, &
              snow=snow, snowncv=snowncv, sr=sr, has_reqc=0, has_reqi=0, has_reqs=0, &
              ids=1, ide=nx, jds=1, jde=ny, kds=1, kde=nz, ims=1, ime=nx, jms=1, jme=ny, kms=1, kme=nz, &
              its=its, ite=ite, jts=jts, jte=jte, kts=kts, kte=kte)
  end subroutine

  !> wsm6init (mp_wsm6.f90:1432-1506) as mp_driver.f90:100 calls it; out(1:60) = the module constants it derives, in the order of
  !! their SAVE declaration (mp_wsm6.f90:44-58)
  subroutine ref_wsm6_init(out) bind(C, name="ref_wsm6_init")
    real(c_float), intent(out) :: out(60)
    call wsm6init(wc_rhoair0, wc_rhowater, wc_rhosnow, wc_cliq, wc_cpv)
    out = [ &
           w6_qc0, w6_qck1, w6_bvtr1, w6_bvtr2, w6_bvtr3, w6_bvtr4, w6_g1pbr, w6_g3pbr, w6_g4pbr, w6_g5pbro2, w6_pvtr, &
           w6_eacrr, w6_pacrr, w6_bvtr6, w6_g6pbr, w6_precr1, w6_precr2, w6_roqimax, w6_bvts1, w6_bvts2, w6_bvts3, &
           w6_bvts4, w6_g1pbs, w6_g3pbs, w6_g4pbs, w6_g5pbso2, w6_pvts, w6_pacrs, w6_precs1, w6_precs2, w6_pidn0r, &
           w6_pidn0s, w6_xlv1, w6_pacrc, w6_pi, w6_bvtg1, w6_bvtg2, w6_bvtg3, w6_bvtg4, w6_g1pbg, w6_g3pbg, w6_g4pbg, &
           w6_g5pbgo2, w6_pvtg, w6_pacrg, w6_precg1, w6_precg2, w6_pidn0g, w6_rslopermax, w6_rslopesmax, w6_rslopegmax, &
           w6_rsloperbmax, w6_rslopesbmax, w6_rslopegbmax, w6_rsloper2max, w6_rslopes2max, w6_rslopeg2max, &
           w6_rsloper3max, w6_rslopes3max, w6_rslopeg3max]
  end subroutine

  !> wsm6 (mp_wsm6.f90:62) on a tile exactly as mp_driver.f90:518-550 calls it (snowncv / graupelncv absent).
  !! Arrays (nx,nz,ny) / (nx,ny).
  subroutine ref_wsm6(nx, nz, ny, th, q, qc, qr, qi, qs, qg, den, pii, p, delz, delt, rain, rainncv, sr, snow, graupel, &
                      its, ite, jts, jte, kts, kte) bind(C, name="ref_wsm6")
    integer(c_int), value :: nx, nz, ny, its, ite, jts, jte, kts, kte
    real(c_float), value :: delt
    real(c_float), intent(inout), dimension(nx,nz,ny) :: th, q, qc, qr, qi, qs, qg
    real(c_float), intent(in), dimension(nx,nz,ny) :: den, pii, p, delz
    real(c_float), intent(inout), dimension(nx,ny) :: rain, rainncv, sr, snow, graupel
    call wsm6(q=q, th=th, qc=qc, qi=qi, qr=qr, qs=qs, qg=qg, pii=pii, p=p, delz=delz, den=den, delt=delt, g=gravity, cpd=cp, &
              cpv=wc_cpv, rd=Rd, rv=Rw, t0c=273.15, ep1=EP1, ep2=EP2, qmin=wc_epsilon, XLS=wc_XLS, XLV0=wc_XLV, XLF0=wc_XLF, &
              den0=wc_rhoair0, denr=wc_rhowater, cliq=wc_cliq, cice=wc_cice, psat=wc_psat, rain=rain, rainncv=rainncv, sr=sr, &
              snow=snow, graupel=graupel, &
              ids=1, ide=nx, jds=1, jde=ny, kds=1, kde=nz, ims=1, ime=nx, jms=1, jme=ny, kms=1, kme=nz, &
              its=its, ite=ite, jts=jts, jte=jte, kts=kts, kte=kte)
  end subroutine
end module icar_ref_shim
