/* include/icar_hip.h -- C ABI of libicar_hip.so: the MI355X (gfx950) implementation of ICAR's
 * per-timestep 3-D grid update (advection + column microphysics + wind balance + halo faces).
 *
 * Every entry point replaces one Fortran procedure of the reference (NCAR/icar, cited as
 * src/...:line).  The reference has no C ABI of its own; the seam chosen is its level-2 physics
 * kernels, which already take plain REAL(4) arrays + WRF-style index triplets.
 * The Fortran-2008 iso_c_binding module that binds these (icar_amd/fortran/icar_hip_mod.f90)
 * and the call sites a maintainer edits are shown in INTEGRATION.md.
 *
 * Conventions
 *  - All 3-D fields are REAL(4) in Fortran order X(ims:ime, kms:kme, jms:jme): i (x) fastest,
 *    then k (z), then j (y).  u is staggered (ims:ime+1), v is (jms:jme+1).
 *    2-D accumulators are REAL(8) X(ims:ime, jms:jme) like domain%...%data_2dd.
 *  - Index arguments (its..kte, ids..kde) are in the same index space as the ims..jme given at
 *    context creation (any lower bound), inclusive, exactly as the reference passes them.
 *  - Functions return 0 on success, non-zero on error; icar_hip_last_error() gives the text.
 *    (The reference's convention is `stop`/`error stop`; the Fortran binding maps !=0 to error stop.)
 *  - One context per coarray image / GPU; calls on a context are serialised by the caller
 *    (same as the reference's module-level SAVE state, which the context replaces).
 *  - No CPU fallback exists: without a visible gfx950 device icar_hip_ctx_create fails.
 */
#ifndef ICAR_HIP_H
#define ICAR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct icar_hip_ctx icar_hip_ctx;

/* Field ids mirror the domain_t members the path touches (src/objects/domain_h.f90:35-50,283-293)
 * and the kVARS order used by the advection dispatch (src/physics/adv_mpdata.f90:512-522). */
enum icar_hip_field {
    ICAR_F_WATER_VAPOR = 0,        /* domain%water_vapor%data_3d            */
    ICAR_F_CLOUD_WATER = 1,        /* domain%cloud_water_mass%data_3d       */
    ICAR_F_RAIN = 2,               /* domain%rain_mass%data_3d              */
    ICAR_F_SNOW = 3,               /* domain%snow_mass%data_3d              */
    ICAR_F_POTENTIAL_TEMPERATURE = 4,
    ICAR_F_CLOUD_ICE = 5,          /* domain%cloud_ice_mass%data_3d         */
    ICAR_F_GRAUPEL = 6,            /* domain%graupel_mass%data_3d           */
    ICAR_F_ICE_NUMBER = 7,         /* domain%cloud_ice_number%data_3d       */
    ICAR_F_RAIN_NUMBER = 8,        /* domain%rain_number%data_3d            */
    ICAR_F_SNOW_NUMBER = 9,        /* domain%snow_number%data_3d            */
    ICAR_F_GRAUPEL_NUMBER = 10,    /* domain%graupel_number%data_3d         */
    ICAR_N_ADVECTABLE = 11,
    ICAR_F_U = 11,                 /* domain%u%data_3d   (nx+1, nz, ny)     */
    ICAR_F_V = 12,                 /* domain%v%data_3d   (nx, nz, ny+1)     */
    ICAR_F_W = 13,                 /* domain%w%data_3d                      */
    ICAR_F_PRESSURE = 14,
    ICAR_F_EXNER = 15,
    ICAR_F_DENSITY = 16,
    ICAR_F_DZ_MASS = 17,           /* domain%dz_mass%data_3d                */
    ICAR_F_JACOBIAN = 18,          /* domain%jacobian                       */
    ICAR_F_JACOBIAN_U = 19,        /* (nx+1, nz, ny)                        */
    ICAR_F_JACOBIAN_V = 20,        /* (nx, nz, ny+1)                        */
    ICAR_F_JACOBIAN_W = 21,
    ICAR_F_ADVECTION_DZ = 22,      /* domain%advection_dz                   */
    ICAR_F_PRECIPITATION = 23,     /* domain%accumulated_precipitation%data_2dd  REAL(8) (nx,ny) */
    ICAR_F_SNOWFALL = 24,          /* domain%accumulated_snowfall%data_2dd       REAL(8) (nx,ny) */
    ICAR_F_GRAUPEL_ACC = 25,       /* domain%graupel%data_2dd                    REAL(8) (nx,ny) */
    /* diagnostic_update outputs / inputs (src/main/time_step.f90:49-198) */
    ICAR_F_PRESSURE_INTERFACE = 26,
    ICAR_F_TEMPERATURE = 27,
    ICAR_F_TEMPERATURE_INTERFACE = 28,
    ICAR_F_U_MASS = 29,
    ICAR_F_V_MASS = 30,
    ICAR_F_W_REAL = 31,
    ICAR_F_DZDX = 32,              /* domain%dzdx  (nx+1, nz, ny)           */
    ICAR_F_DZDY = 33,              /* domain%dzdy  (nx, nz, ny+1)           */
    ICAR_F_SURFACE_PRESSURE = 34,  /* domain%surface_pressure%data_2d  REAL(4) (nx,ny) */
    /* linear-theory winds (src/physics/linear_winds.f90:840-1127) */
    ICAR_F_Z = 35,                 /* domain%z%data_3d (mass-level height)  */
    ICAR_F_NSQUARED = 36,          /* domain%nsquared%data_3d               */
    /* optional column integrals of diagnostic_update (src/main/time_step.f90:126-144), REAL(4) (nx,ny) */
    ICAR_F_IVT = 37,               /* domain%ivt%data_2d  integrated vapour transport            */
    ICAR_F_IWV = 38,               /* domain%iwv%data_2d  integrated water vapour                */
    ICAR_F_IWL = 39,               /* domain%iwl%data_2d  integrated liquid (cloud + rain)       */
    ICAR_F_IWI = 40,               /* domain%iwi%data_2d  integrated ice (ice + snow + graupel)  */
    /* update_winds, windtype kCONSERVE_MASS (src/physics/wind.f90:301-306): the host uploads domain%zr_u / zr_v, or
     * zfr_u / zfr_v when options%parameters%use_terrain_difference */
    ICAR_F_ZR_U = 41,              /* (nx+1, nz, ny)                        */
    ICAR_F_ZR_V = 42,              /* (nx, nz, ny+1)                        */
    /* make_winds_grid_relative (src/physics/wind.f90:236-287): domain%sintheta / costheta, REAL(8) (nx,ny) */
    ICAR_F_SINTHETA = 43,
    ICAR_F_COSTHETA = 44,
    ICAR_N_FIELDS = 45
};

enum { ICAR_ADV_UPWIND = 1, ICAR_ADV_MPDATA = 2 };   /* kADV_UPWIND / kADV_MPDATA, icar_constants.f90:341 */

/* ---- context + device-resident field mirrors (replaces module SAVE state) ---- */
int icar_hip_ctx_create(icar_hip_ctx **ctx, int device,
                        int ims, int ime, int kms, int kme, int jms, int jme);
int icar_hip_ctx_destroy(icar_hip_ctx *ctx);
/* Run all kernels of this context on the caller's HIP stream (hipStream_t); NULL = own stream. */
int icar_hip_set_stream(icar_hip_ctx *ctx, void *hip_stream);
int icar_hip_synchronize(icar_hip_ctx *ctx);
/* Second HIP stream of the context (created on first use).  time_step.f90:512-526 orders
 *     mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve
 * so that the interior microphysics overlaps the coarray PUTs; here the strips, the pack kernels and the exchange stay
 * on the main stream while the interior launch runs beside them on the aux stream:
 *     aux_fork   aux waits for everything issued on the main stream so far
 *     aux_begin  ... entry points called here launch on the aux stream ...  aux_end
 *     aux_join   the main stream waits for everything issued on the aux stream so far
 * Work issued between fork and join on the two streams must touch disjoint cells (strips vs interior columns). */
int icar_hip_aux_fork(icar_hip_ctx *ctx);
int icar_hip_aux_begin(icar_hip_ctx *ctx);
int icar_hip_aux_end(icar_hip_ctx *ctx);
int icar_hip_aux_join(icar_hip_ctx *ctx);
/* Lazily allocates the device mirror of a field.  Host buffers are the domain_t arrays
 * (contiguous, Fortran order).  Element count: icar_hip_field_count(). */
int icar_hip_field_upload(icar_hip_ctx *ctx, int field, const void *host);
int icar_hip_field_download(icar_hip_ctx *ctx, int field, void *host);
int icar_hip_field_fill(icar_hip_ctx *ctx, int field, double value);
/* Current device pointer of a field (advect() ping-pongs the advected scalars, so re-query
 * after every icar_hip_advect call). */
int icar_hip_field_device_ptr(icar_hip_ctx *ctx, int field, void **dptr);
size_t icar_hip_field_count(const icar_hip_ctx *ctx, int field);
size_t icar_hip_field_elem_size(int field);

/* ---- A1: Courant-number winds U_m,V_m,W_m -----------------------------------------------------
 * replaces setup_module_winds (src/physics/advect.f90:306-351, scheme=1) and the inline block of
 * mpdata (src/physics/adv_mpdata.f90:496-506, scheme=2; the two differ in multiplication order). */
int icar_hip_setup_winds(icar_hip_ctx *ctx, int scheme, float dt, float dx, int advect_density);

/* ---- A2-A5: advect the listed scalars with the winds of the last icar_hip_setup_winds --------
 * replaces the per-variable advect3d dispatch of upwind (src/physics/advect.f90:380-418) and
 * mpdata (src/physics/adv_mpdata.f90:463-524; advect3d :356-418, upwind_advection :44-105,
 * mpdata_fluxes :107-255, flux_limiter :257-354 + adv_mpdata_FCT_core.f90:47-116).
 * fields[] are ICAR_F_* ids < ICAR_N_ADVECTABLE (the caller derives them from
 * options%vars_to_advect); mpdata_order / fct = options%adv_options.
 * Call icar_hip_setup_winds again whenever u, v, w, density or a jacobian changed on the device (apply_forcing,
 * balance_uvw, the wind solvers): the call fails if the Courant winds are known to be stale.
 * Arithmetic: the upwind scheme (and mpdata_order 1) is bit-identical to the reference; MPDATA's corrective iterations
 * use 1-ulp reciprocals and agree with it to 1e-5 of the local field scale (tests/test_gpu_advect.py). */
int icar_hip_advect(icar_hip_ctx *ctx, int scheme, int mpdata_order, int fct, int advect_density,
                    const int *fields, int nfields);

/* Corrective iterations of MPDATA in the reference's own operation order (on = 1): donor cell, mpdata_fluxes, flux_limiter +
 * FCT core and the final donor cell as separate launches per scalar, IEEE division, no contraction -- the advected fields are
 * then BIT-IDENTICAL to src/physics/adv_mpdata.f90:356-418 on the CPU, and so is a whole sub-step sequence
 * (tests/test_gpu_advect.py, tests/test_gpu_trajectory.py).  Costs ~100 B of HBM traffic per scalar-cell instead of 8; the
 * default (on = 0) is the fused kernel.  No counterpart in the reference (it has one arithmetic). */
int icar_hip_mpdata_exact(icar_hip_ctx *ctx, int on);
/* 1 while the Courant winds of the last icar_hip_setup_winds still belong to the state (nothing has rewritten u, v, w, density
 * or a jacobian since), else 0: a host that launches the setup early -- beside the interior microphysics, see INTEGRATION.md --
 * asks before it skips the setup in advect(). */
int icar_hip_winds_valid(icar_hip_ctx *ctx);

/* ---- M1: mp_simple_driver (src/physics/mp_simple.f90:595-646) on the tile its..kte -----------
 * uses PRESSURE, POTENTIAL_TEMPERATURE, EXNER, DENSITY, WATER_VAPOR, CLOUD_WATER, RAIN, SNOW,
 * DZ_MASS; adds the tile's surface fluxes to PRECIPITATION / SNOWFALL exactly as
 * process_subdomain does (src/physics/mp_driver.f90:587-595).  *err_count (may be NULL) receives
 * the number of columns on which the reference would have hit its `stop` in phase_change. */
int icar_hip_mp_simple(icar_hip_ctx *ctx, float dt,
                       int its, int ite, int jts, int jte, int kts, int kte, int *err_count);
/* process_halo's strips (src/physics/mp_driver.f90:609-658: one process_subdomain -> mp_simple_driver call per strip) as
 * ONE launch over up to 4 non-overlapping tiles {its,ite,jts,jte} (as icar_hip_mp_tiles returns them). */
int icar_hip_mp_simple_tiles(icar_hip_ctx *ctx, float dt, int ntiles, const int tiles[][4], int kts, int kte, int *err_count);

/* ---- M2-M4: Thompson microphysics ------------------------------------------------------------
 * thompson_init (src/physics/mp_thompson.f90:342-766; params/flags = mp_options_type in the
 * order Nt_c,TNO,am_s,rho_g,av_s,bv_s,fv_s,av_g,bv_g,av_i,Ef_si,Ef_rs,Ef_rg,Ef_ri,C_cubes,
 * C_sqrd,mu_r,t_adjust ; Ef_rw_l,Ef_sw_l) and mp_gt_driver (:772-1044). */
int icar_hip_thompson_init(icar_hip_ctx *ctx, const float params[18], const int flags[2]);
int icar_hip_thompson(icar_hip_ctx *ctx, float dt,
                      int its, int ite, int jts, int jte, int kts, int kte,
                      int ids, int ide, int jds, int jde, int kds, int kde);

/* process_halo (src/physics/mp_driver.f90:609-658) calls process_subdomain -> mp_gt_driver once per strip; the four
 * one-cell strips are latency-bound when launched one after another, so this entry runs up to 4 tiles
 * {its,ite,jts,jte} (as icar_hip_mp_tiles returns them) in ONE launch.  Tiles must not overlap. */
int icar_hip_thompson_tiles(icar_hip_ctx *ctx, float dt, int ntiles, const int tiles[][4], int kts, int kte,
                            int ids, int ide, int jds, int jde, int kds, int kde);
/* Download one Thompson lookup table by its reference name (tcg_racg ... t_Efsw, Fortran order) for
 * cross-checks against ICAR's own qr_acr_qg_mpt.dat / qr_acr_qs_mpt.dat / freezeH2O_mpt.dat caches
 * (src/physics/mp_thompson.f90:2870-2887).  out may be NULL to query the element count. */
int icar_hip_thompson_table(icar_hip_ctx *ctx, const char *name, double *out, size_t capacity, size_t *count);

/* ---- M0: tile bookkeeping of mp()/process_halo (src/physics/mp_driver.f90:609-772) -----------
 * Fills tiles[n][4] = {its,ite,jts,jte} for halo>0 (W,E,S,N strips; corners once) or for the
 * interior shrunk by subset; returns the number of tiles (integer-exact restatement). */
int icar_hip_mp_tiles(int its, int ite, int jts, int jte, int halo, int subset, int tiles[4][4]);

/* ---- WSM3 (src/physics/mp_wsm3.f90), the kMP_WSM3 slot of mp()'s dispatch (mp_driver.f90:103-106, :552-585) -------------
 * wsm3_init == wsm3init(rhoair0, rhowater, rhosnow, cliq, cpv).  wsm3 == process_subdomain's call: qci = CLOUD_WATER,
 * qrs = RAIN, w = W_REAL, den = DENSITY, pii = EXNER, delz = DZ_MASS; precipitation / snowfall of the call are added to the
 * REAL(8) ICAR_F_PRECIPITATION / ICAR_F_SNOWFALL (:587-595).  Tile bounds 1-based inclusive like its..kte. */
int icar_hip_wsm3_init(icar_hip_ctx *ctx);
int icar_hip_wsm3(icar_hip_ctx *ctx, float dt, int its, int ite, int jts, int jte, int kts, int kte);

/* ---- WSM6 (src/physics/mp_wsm6.f90), the kMP_WSM6 slot of mp()'s dispatch (mp_driver.f90:98-101, :518-550) ----------------
 * wsm6_init == wsm6init(rhoair0, rhowater, rhosnow, cliq, cpv) (:1432-1506).  wsm6 == process_subdomain's call of wsm6 (:62):
 * q = WATER_VAPOR, qc = CLOUD_WATER, qi = CLOUD_ICE, qr = RAIN, qs = SNOW, qg = GRAUPEL, th = POTENTIAL_TEMPERATURE, with
 * EXNER, PRESSURE, DZ_MASS, DENSITY; the tile's rain / snow / graupel surface sums are added to PRECIPITATION, SNOWFALL and
 * GRAUPEL_ACC as mp_driver.f90:587-595 does.  4..64 levels. */
int icar_hip_wsm6_init(icar_hip_ctx *ctx);
int icar_hip_wsm6(icar_hip_ctx *ctx, float dt, int its, int ite, int jts, int jte, int kts, int kte);
/* process_halo's strips (src/physics/mp_driver.f90:609-658) in ONE sequence of launches over up to 4 non-overlapping tiles
 * {its,ite,jts,jte} (as icar_hip_mp_tiles returns them), like icar_hip_thompson_tiles / icar_hip_mp_simple_tiles */
int icar_hip_wsm6_tiles(icar_hip_ctx *ctx, float dt, int ntiles, const int tiles[][4], int kts, int kte);

/* ---- T2: CFL reduction for compute_dt (src/main/time_step.f90:217-330, cfl_strictness 3) -----
 * out = max over the tile of max(|u_i|,|u_i+1|)/dx + max(|v_j|,|v_j+1|)/dx + max(|w_k|,|w_k-1|)/dz_levels(k) */
int icar_hip_max_courant(icar_hip_ctx *ctx, float dx, const float *dz_levels, float *out);
/* Same reduction, result left in DEVICE memory at d_out (one REAL(4) owned by the caller), no host synchronisation:
 * `call co_min(seconds)` (time_step.f90:413) becomes a 1-element all-reduce(max) of d_out over the images on the device
 * (dt = cfl_reduction_factor / max is monotone, so the minimum dt is the quotient of the maximum). */
int icar_hip_max_courant_device(icar_hip_ctx *ctx, float dx, const float *dz_levels, void *d_out);
/* the same reduction taken ahead of time, on the current stream (typically the second one, beside the advection, right after the
 * forcing of u, v, w): the next icar_hip_max_courant / _device call with the same arguments returns this value without launching
 * anything, provided no entry point has written u, v or w in between (the library counts those writes; handing out a raw
 * device pointer to a wind field disables the shortcut for good). */
int icar_hip_max_courant_prefetch(icar_hip_ctx *ctx, float dx, const float *dz_levels);
/* the other cfl_strictness settings (:238-259, :293-305) also need out[0..2] = maxval(abs(u)), maxval(abs(v)), maxval(abs(w)) */
int icar_hip_max_abs_winds(icar_hip_ctx *ctx, float out[3]);

/* ---- T3: diagnostic_update (src/main/time_step.f90:49-198) ------------------------------------
 * exner=(p/1e5)^(Rd/cp), interface pressure/temperature, surface pressure, T=theta*exner,
 * rho=p/(Rd T), u_mass, v_mass, w_real (uses DZDX, DZDY, JACOBIAN).  The optional column integrals ivt / iwv / iwl / iwi
 * (compute_ivt, compute_iq: src/utilities/atm_utilities.f90:35-102) are computed for every one of ICAR_F_IVT..IWI the
 * host has uploaded once (= `associated(domain%ivt%data_2d)`); iwl / iwi sum the hydrometeor fields that are on the
 * device, like the reference's `associated` tests.  The 10 m winds need roughness_z0 (LSM) and stay host-side. */
int icar_hip_diagnostic_update(icar_hip_ctx *ctx);
/* the same in two parts, for a host that overlaps: parts = 1: everything but w_real (what the microphysics and the advection
 * read: exner, density ...) ; 2: w_real only (:165-194; read by WSM3 and the output, not by Thompson / mp_simple / WSM6 / advect:
 * a streaming kernel that can run beside the VALU-bound interior microphysics) ; 3: both = icar_hip_diagnostic_update. */
int icar_hip_diagnostic_update_parts(icar_hip_ctx *ctx, int parts);

/* ---- F1: apply_forcing / enforce_limits (src/objects/domain_obj.f90:2383-2448, 2228-2243) ----
 * dqdt mirrors variable_t%dqdt_3d.  For each listed field: force_boundaries[i]!=0 -> only the true
 * domain edges named by the west/east/south/north flags (W/E columns without corners, S/N full
 * rows) get  x += dqdt*dt ; otherwise the whole field does (u, v, w, pressure ...).  dt is REAL(8)
 * like time_delta_t%seconds(). */
int icar_hip_dqdt_upload(icar_hip_ctx *ctx, int field, const void *host);
int icar_hip_dqdt_download(icar_hip_ctx *ctx, int field, void *host);
int icar_hip_apply_forcing(icar_hip_ctx *ctx, double dt_seconds, const int *fields, const int *force_boundaries, int nfields,
                           int west_boundary, int east_boundary, int south_boundary, int north_boundary);
int icar_hip_enforce_limits(icar_hip_ctx *ctx, const int *fields, int nfields);

/* ---- W1: balance_uvw (src/physics/wind.f90:81-169): w from the horizontal divergence ---------- */
int icar_hip_balance_uvw(icar_hip_ctx *ctx, float dx);
/* same on u/v/w%meta_data%dqdt_3d -- the form update_winds uses at every forcing step after the first
 * (src/physics/wind.f90:341-360); the w tendency mirror is created if needed. */
int icar_hip_balance_uvw_update(icar_hip_ctx *ctx, float dx);

/* ---- make_winds_grid_relative (src/physics/wind.f90:236-287): rotate the forcing winds from E-W / N-S to the grid's
 * orientation -- destagger u, v to the mass grid, rotate by ICAR_F_SINTHETA / ICAR_F_COSTHETA (REAL(8), (nx,ny)),
 * restagger, extrapolate the two lost edge cells.  update_winds calls it on both of its branches (:300 on u, v; :338 on
 * their dqdt_3d when update != 0).  Not the identity even for an unrotated grid (the double averaging smooths). */
int icar_hip_make_winds_grid_relative(icar_hip_ctx *ctx, int update);

/* ---- iterative_winds (src/physics/wind.f90:371-498), SURVEY 8(f) rank 4 -----------------------
 * The host keeps the reference's control flow: [exchange_u, exchange_v], balance_uvw, correct_w, then
 * wind_iterations+1 times { sweep(1); exchange_u; exchange_v }.  On one image the exchanges are no-ops and
 * sweep(wind_iterations+1) runs the whole loop.  update != 0 works on u/v/w%meta_data%dqdt_3d (wind.f90:392-401).
 *   correct_w  :430-441  w(i,k,j) -= min(sum(dz(1:k))/sum(dz), 1) * w(i,kme,j)
 *   sweep      :455-481  div = calc_divergence(u,v,w) (:172-228); ADJ = div/(-2/dx); u, v faces +-ADJ*0.5 */
int icar_hip_iterative_winds_correct_w(icar_hip_ctx *ctx, int update);
int icar_hip_iterative_winds_sweep(icar_hip_ctx *ctx, float dx, int nsweeps, int update);

/* mass_conservative_acceleration (src/physics/wind.f90:500-511): u = u / ICAR_F_ZR_U, v = v / ICAR_F_ZR_V on the winds
 * (update == 0) or on their meta_data%dqdt_3d (update != 0), as update_winds does for windtype kCONSERVE_MASS */
int icar_hip_mass_conservative_acceleration(icar_hip_ctx *ctx, int update);

/* ---- update_winds (src/physics/wind.f90:289-369) in ONE call ------------------------------------
 * make_winds_grid_relative -> [linear_perturb (windtype 1, 5)] -> [mass_conservative_acceleration (2)] ->
 * [iterative_winds (3, 5): exchange_u / exchange_v, balance_uvw, correct_w, wind_iterations+1 x { sweep; exchange_u;
 * exchange_v } (:371-498)] -> balance_uvw, on u, v, w (update = 0: the first call of a run) or on their
 * meta_data%dqdt_3d (update = 1: every later forcing step); update < 0 lets the context decide like the reference's
 * `if (.not.allocated(domain%sintheta))`.  The staggered exchanges use the context's communicator (icar_hip_comm_init /
 * _init_host), so a multi-image host gets the same winds as a single-image one on the cells it owns.  dx = domain%dx,
 * halo = grid%halo_size, wind_iterations = options%parameters%wind_iterations.  Needs what the parts need: SINTHETA /
 * COSTHETA uploaded (init_winds, :512-590), the linear-wind LUT built for windtype 1 / 5, ZR_U / ZR_V for 2. */
int icar_hip_update_winds(icar_hip_ctx *ctx, int windtype, int wind_iterations, float dx, int halo, int update);
/* `call domain%u%exchange_u(); call domain%v%exchange_v()` (src/objects/exchangeable_obj.f90:158-229) on u, v
 * (update = 0) or their dqdt_3d (1): one message per neighbour through the context's communicator.  Collective. */
int icar_hip_exchange_uv(icar_hip_ctx *ctx, int halo, int update);

/* ---- W3: linear-theory wind look-up table (src/physics/linear_winds.f90) ----------------------
 * options%lt_options (src/objects/options_obj.f90:1400-1530; defaults there: buffer 50, stability_window_size 10,
 * vert_smooth 10, max/min_stability 6e-4/1e-7, N_squared 3e-5, linear_contribution 1, linear_update_fraction 0.2,
 * dir 0..2pi x24, spd 0..30 x6, nsq log(min)..log(max) x5, minimum_layer_size 100). */
typedef struct icar_hip_lt_options {
    int buffer, stability_window_size, vert_smooth;
    int variable_N, smooth_nsq;
    float max_stability, min_stability, N_squared, linear_contribution, linear_update_fraction;
    float dirmax, dirmin, spdmax, spdmin, nsqmax, nsqmin;
    int n_dir_values, n_nsq_values, n_spd_values;
    float minimum_layer_size;
} icar_hip_lt_options;

/* setup_linwinds (:1180-1225): buffered + edge-blended + smoothed terrain (add_buffer_topo :351-418, twice),
 * forward 2-D FFT / (nx*ny), fftshift (single-precision temp, src/utilities/fftshift.f90:95-117), the wavenumber
 * axes of initialize_linear_theory_data (:426-470), the LUT axes (:648-650) and zeroed hi_[uv]_perturbation.
 * global_terrain is domain%global_terrain (nx_global, ny_global) Fortran order; ids/jds = global index of its first
 * cell in the index space of the context's ims/jms (1 in the reference). */
int icar_hip_linwinds_setup(icar_hip_ctx *ctx, const icar_hip_lt_options *opt, const float *global_terrain,
                            int nx_global, int ny_global, int ids, int jds, float dx);
/* domain%terrain_frequency as (re,im) pairs, (fftnx, fftny) Fortran order; out may be NULL to query the size. */
int icar_hip_linwinds_terrain_frequency(icar_hip_ctx *ctx, double *out, size_t capacity_complex, int *fftnx, int *fftny);
/* linear_perturbation (constant-z form, :239-276 -> linear_perturbation_at_height :181-237): real parts of
 * lt_data%u_perturb / v_perturb on the (fftnx, fftny) grid, host buffers. */
int icar_hip_linear_perturbation(icar_hip_ctx *ctx, float U, float V, float Nsq, float z_bottom, float z_top,
                                 float minimum_step, double *u_perturb, double *v_perturb);
/* initialize_spatial_winds (:596-830), constant-z branch: every (dir, spd, N^2) x level entry for this context's
 * tile; z_bottom/z_top[nz] = layer_height -/+ dz_levels/2 (:751-753). */
int icar_hip_linwinds_build_lut(icar_hip_ctx *ctx, const float *z_bottom, const float *z_top, int nz);
/* same, options%parameters%space_varying_dz branch (:738-748 -> linear_perturbation_varyingz :280-344):
 * z_bottom = global_z_interface - global_terrain, z_top = z_bottom + global_dz_interface, both
 * (nx_global, nz, ny_global) Fortran order. */
int icar_hip_linwinds_build_lut_varying(icar_hip_ctx *ctx, const float *z_bottom, const float *z_top, int nz);
/* hi_u_LUT / hi_v_LUT in the reference's index order (n_spd, n_dir, n_nsq, nx[+1], nz, ny[+1]) -- the order of the
 * disk cache src/io/lt_lut_io.f90 -- component 0 = u, 1 = v.  upload replaces read_LUT (:659). */
int icar_hip_linwinds_lut_download(icar_hip_ctx *ctx, int component, float *host);
int icar_hip_linwinds_lut_upload(icar_hip_ctx *ctx, int component, const float *host);
/* one entry hi_u_LUT(spd, dir, nsq, :, :, :) / hi_v_LUT(...) (0-based indices) as a field of the tile's u / v shape (Fortran
 * order (nx[+1], nz, ny[+1])): a production LUT is tens of GB, an entry 20-40 MB */
int icar_hip_linwinds_lut_entry(icar_hip_ctx *ctx, int component, int spd, int dir, int nsq, float *host);
/* hi_u_perturbation / hi_v_perturbation (the relaxed perturbation state, :1263-1268), for restarts and tests. */
int icar_hip_linwinds_perturbation_download(icar_hip_ctx *ctx, int component, float *host);
int icar_hip_linwinds_perturbation_upload(icar_hip_ctx *ctx, int component, const float *host);

/* ---- W2: spatial_winds (:840-1127, reverse=.false.) -------------------------------------------
 * N^2 (calc_stability, src/utilities/atm_utilities.f90:401-467) -> clamp -> log -> vertical (+-vert_smooth) and
 * horizontal (smooth_array ydim=3, src/utilities/array_utilities.f90:308-417) smoothing -> per cell bracket search
 * and 8-corner interpolation of the LUTs -> relaxation of hi_[uv]_perturbation -> added to U/V (update=0) or to their
 * dqdt_3d mirrors (update=1).  Reads POTENTIAL_TEMPERATURE, EXNER, Z, WATER_VAPOR and whichever of CLOUD_WATER,
 * CLOUD_ICE, RAIN, SNOW are on the device ("associated"); writes NSQUARED. */
int icar_hip_spatial_winds(icar_hip_ctx *ctx, int update);

/* ---- H1: halo faces (src/objects/exchangeable_obj.f90:138-356) --------------------------------
 * dir: 0=north 1=south 2=east 3=west.  pack gathers what exchangeable%put_<dir> would PUT
 * (my interior planes next to that edge, width halo, N/S over the full memory width so corners
 * travel) for all listed fields into one contiguous device buffer; unpack scatters a received
 * buffer into my halo planes like retrieve_<dir>_halo.  Buffers are device pointers; the element
 * count per field is icar_hip_halo_count(). */
size_t icar_hip_halo_count(const icar_hip_ctx *ctx, int dir, int halo);
/* Staggered faces: exchange_u / exchange_v (exchangeable_obj.f90:158-229) move halo+1 planes in the staggered
 * direction, which the cell-grid packers above cannot express.  box_pack gathers the 0-based box
 * [i0, i0+ni) x all levels x [j0, j0+nj) of one REAL(4) 3-D field (which = 0: data_3d, 1: meta_data%dqdt_3d) into a
 * device buffer laid out [nj][nz][ni]; box_unpack scatters it back.  icar_amd/halo.py holds the index table. */
int icar_hip_box_pack(icar_hip_ctx *ctx, int field, int which, int i0, int ni, int j0, int nj, void *dbuf);
int icar_hip_box_unpack(icar_hip_ctx *ctx, int field, int which, int i0, int ni, int j0, int nj, const void *dbuf);
int icar_hip_halo_pack(icar_hip_ctx *ctx, int dir, int halo, const int *fields, int nfields, void *dbuf);
int icar_hip_halo_unpack(icar_hip_ctx *ctx, int dir, int halo, const int *fields, int nfields, const void *dbuf);
/* the same for several directions in ONE launch (dirs[z], dbufs[z], z < ndirs <= 4): what domain%halo_send /
 * halo_retrieve (objects/domain_obj.f90:109-143) issue per step -- 2 launches instead of 8 for an image with 4 neighbours */
int icar_hip_halo_pack_dirs(icar_hip_ctx *ctx, int ndirs, const int *dirs, int halo, const int *fields, int nfields, void *const *dbufs);
int icar_hip_halo_unpack_dirs(icar_hip_ctx *ctx, int ndirs, const int *dirs, int halo, const int *fields, int nfields, void *const *dbufs);

/* ---- H1 transport + co_min: domain%halo_send / halo_retrieve (src/objects/domain_obj.f90:109-143 over
 * src/objects/exchangeable_obj.f90:138-356) and `call co_min(seconds)` (src/main/time_step.f90:413) -------------------
 * One message per neighbour carries every exchanged scalar (the reference PUTs once per variable and direction).
 * neighbors[dir] (dir 0=north 1=south 2=east 3=west) is the 0-based rank of the image on that side (= image - 1 of
 * grid_t's neighbour), ICAR_NEIGHBOR_NONE at a domain boundary, or ICAR_NEIGHBOR_SELF for an edge that wraps around to the
 * tile's own opposite edge (periodic single image, what src/tests/test_mpdata.f90 does by hand; both edges of an axis).
 *   icar_hip_comm_init       RCCL (ncclSend / ncclRecv over xGMI on the context's stream, nothing waits on the host).
 *                            uid = the 128 bytes image 1 got from icar_hip_comm_unique_id and broadcast (co_broadcast in a
 *                            coarray host); uid == NULL is allowed for one image without peers.
 *                            One GPU per image: RCCL refuses two ranks on one device.
 *   icar_hip_comm_init_host  the same entry points with the messages staged through POSIX shared memory `shm_name`
 *                            (unique per run; slot_bytes >= the largest message = max halo_count * number of exchanged
 *                            fields * 4, the same on every image): for boxes with fewer GPUs than images.  A functional
 *                            path, not a fast one.
 * halo_send packs (one launch) and posts the transfers; halo_retrieve is the `sync images` + unpack (one launch, the
 * reference's N, S, E, W precedence at the corner cells).  Exactly one halo_retrieve per halo_send, same arguments. */
enum { ICAR_NEIGHBOR_NONE = -1, ICAR_NEIGHBOR_SELF = -2 };
enum { ICAR_COMM_NONE = 0, ICAR_COMM_LOCAL = 1, ICAR_COMM_RCCL = 2, ICAR_COMM_HOST = 3 };
int icar_hip_comm_unique_id(char uid[128]);
int icar_hip_comm_init(icar_hip_ctx *ctx, int nranks, int rank, const char uid[128], const int neighbors[4]);
int icar_hip_comm_init_host(icar_hip_ctx *ctx, int nranks, int rank, const char *shm_name, size_t slot_bytes, const int neighbors[4]);
int icar_hip_comm_destroy(icar_hip_ctx *ctx);
int icar_hip_comm_kind(icar_hip_ctx *ctx);                      /* ICAR_COMM_* */
/* how long the host-staged transport waits for a neighbouring image before it reports an error (default 60 s; a debugger or
 * a long host pause on a neighbour wants more).  Process-wide. */
int icar_hip_comm_timeout(double seconds);
/* num_images() as the transport itself reports it (ncclCommCount / the shared segment's header; 1 without a transport) */
int icar_hip_comm_ranks(icar_hip_ctx *ctx, int *nranks);
/* A self-test of exchangeable_t%send / %retrieve (src/objects/exchangeable_obj.f90:138-356) on this communicator: one
 * exchange of a field stamped with this image's number, verified on the device -- every halo cell must carry the stamp of
 * the neighbour on its side (corner cells aside).  The field used (water vapour) is restored.  n_bad = offending cells.
 * Collective: every image of the communicator calls it. */
int icar_hip_halo_selfcheck(icar_hip_ctx *ctx, int halo, int *n_bad);
int icar_hip_halo_send(icar_hip_ctx *ctx, int halo, const int *fields, int nfields);
int icar_hip_halo_retrieve(icar_hip_ctx *ctx, int halo, const int *fields, int nfields);
/* in/out: *value becomes the minimum (maximum) over the images; one image: unchanged */
int icar_hip_co_min(icar_hip_ctx *ctx, double *value);
int icar_hip_co_max(icar_hip_ctx *ctx, double *value);

/* ---- T1 / T2 / M0: the sub-step loop itself (src/main/time_step.f90:440-551) ----------------------------------------
 * icar_hip_step_configure hands the library the members of options_t / grid_t the loop reads; after that
 *   icar_hip_compute_dt  == compute_dt (:217-330) on this image's tile
 *   icar_hip_update_dt   == update_dt (:375-423): compute_dt (:217-330, every cfl_strictness) + co_min + the 120 s cap;
 *                           fails with "ERROR time step too small" where the reference stops (:322-328)
 *   icar_hip_mp          == mp(domain, options, dt, halo, subset) (src/physics/mp_driver.f90:673-772) incl. the
 *                           update_interval gating (:698-713); halo / subset < 0 = argument not present
 *   icar_hip_advect_step == advect(domain, options, dt) (src/physics/advection_driver.f90:51-77)
 *   icar_hip_substep     == one pass of :474-539: diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1) ->
 *                           halo_retrieve -> advect -> apply_forcing [-> enforce_limits], with the interior microphysics and
 *                           the streaming kernels issued on the context's second stream beside the heavy ones
 *   icar_hip_step        == step(domain, end_time, options) (:440-551); the model clock lives in the context.
 *                           An error from icar_hip_step / icar_hip_step_n leaves the fields undefined (the reference STOPs
 *                           where update_dt fails, :322-328): when the failing sub-step had already been opened the context
 *                           is marked failed and every stepping entry point returns an error until the caller has reloaded
 *                           the fields and set the clock again (icar_hip_model_time_set)
 * The library keeps the model clock (domain%model_time) and mp_driver.f90's SAVE variable last_model_time. */
typedef struct icar_hip_step_config {
    int advection;                  /* options%physics%advection: 0, ICAR_ADV_UPWIND, ICAR_ADV_MPDATA            */
    int microphysics;               /* options%physics%microphysics: 0, 1 Thompson, 2 mp_simple, 4 WSM6, 6 WSM3  */
    int mpdata_order;               /* options%adv_options%mpdata_order                                           */
    int flux_corrected_transport;   /* options%adv_options%flux_corrected_transport                               */
    int advect_density;             /* options%parameters%advect_density                                          */
    int cfl_strictness;             /* options%parameters%cfl_strictness (1..5)                                   */
    float cfl_reduction_factor;     /* options%parameters%cfl_reduction_factor                                    */
    float dx;                       /* domain%dx                                                                  */
    float mp_update_interval;       /* options%mp_options%update_interval                                         */
    int top_mp_level;               /* options%mp_options%top_mp_level                                            */
    int halo_size;                  /* grid%halo_size                                                             */
    int its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde;                     /* grid_t                 */
    int west_boundary, east_boundary, south_boundary, north_boundary;                   /* grid_t                 */
    int diagnostics;                /* 1: diagnostic_update at the top of the sub-step (:474), as the reference   */
    int prefetch_dt;                /* 1: take the CFL reduction of the next update_dt beside the advection       */
    int n_advect, advect_fields[ICAR_N_ADVECTABLE];      /* options%vars_to_advect in the dispatch order of adv_mpdata.f90:512-522 */
    int n_exchange, exchange_fields[ICAR_N_ADVECTABLE];  /* the exchangeable members, halo_send order               */
    int n_forced, forced_fields[16], force_boundaries[16];   /* apply_forcing's variables (domain_obj.f90:2383-2448) */
} icar_hip_step_config;
int icar_hip_step_configure(icar_hip_ctx *ctx, const icar_hip_step_config *cfg, const float *dz_levels);
int icar_hip_model_time_set(icar_hip_ctx *ctx, double seconds);
double icar_hip_model_time(const icar_hip_ctx *ctx);
int icar_hip_mp_reset(icar_hip_ctx *ctx);                       /* mp_init / mp_finish: last_model_time = -999 */
int icar_hip_compute_dt(icar_hip_ctx *ctx, double *dt_seconds);   /* compute_dt alone: this image, no co_min, no cap */
int icar_hip_update_dt(icar_hip_ctx *ctx, double *dt_seconds);
int icar_hip_mp(icar_hip_ctx *ctx, double dt, int halo, int subset);
int icar_hip_advect_step(icar_hip_ctx *ctx, double dt);
int icar_hip_substep(icar_hip_ctx *ctx, double dt_seconds, int enforce_limits);
int icar_hip_step(icar_hip_ctx *ctx, double end_time_seconds, int *nsteps);
/* the same loop for a given NUMBER of sub-steps (update_dt -> substep -> clock += dt each), no end-of-interval clamp and no
 * enforce_limits: what a benchmark times as "K passes of the hot path".  dt_last (may be NULL) receives the last dt. */
int icar_hip_step_n(icar_hip_ctx *ctx, int nsteps, double *dt_last);

/* ---- measurement helpers --------------------------------------------------------------------- */
/* Average duration (ms) of the launches of a named kernel group since the last reset, measured
 * with HIP events on the context's stream (bench.py roofline block). group: "advect", "mp". */
int icar_hip_timing_enable(icar_hip_ctx *ctx, int on);
/* restrict the timers to a comma-separated list of groups ("advect", "advect,mp,winds"; NULL or "" = all): every timed
 * scope costs its stream two timestamped barrier packets, ~5 us -- a dozen groups per sub-step are 10 % of a small tile's step */
int icar_hip_timing_groups(icar_hip_ctx *ctx, const char *groups_csv);
int icar_hip_timing_read(icar_hip_ctx *ctx, const char *group, double *total_ms, int *launches);
int icar_hip_timing_reset(icar_hip_ctx *ctx);

const char *icar_hip_last_error(void);
const char *icar_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ICAR_HIP_H */
