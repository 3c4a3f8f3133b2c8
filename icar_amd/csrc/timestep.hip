// icar_amd/csrc/timestep.hip -- rows T1 / T2 / M0 on the library side of the C ABI: the sub-step loop of
// src/main/time_step.f90:440-551 (step), :375-423 (update_dt), :217-330 (compute_dt) and the tile bookkeeping of
// src/physics/mp_driver.f90:673-772 (mp) issued from ONE entry point each, with the two-stream choreography inside the
// library, so that a Fortran host gets exactly what the Python host gets and a small tile is not bound by per-launch host
// overhead (12-25 ctypes / iso_c_binding round trips per sub-step before).
//
// What runs beside what (same launches, operands and results as the plain sequence):
//   main stream (the critical path)                       second stream (side work, done before each join)
//   diagnostic_update part 1 (exner, T, rho, ...)
//   mp(subset=1) interior                                 mp(halo=1) strips -> halo_send (pack + RCCL)   time_step.f90:512-526
//   |                                                     setup_module_winds (+ MPDATA coefficients) of the advect() that follows
//   halo_retrieve (unpack)  <----------------------------- join
//   advect                                                w_real diagnostic, forcing of u, v, w, p, CFL reduction of the next step
//   forcing of the advected scalars (boundary ring) <----- join
//   enforce_limits (last two sub-steps)
#include "ctx.h"
#include <chrono>
#include "comm.h"
#include <cmath>
#include <cstring>

// physics selectors, src/constants/icar_constants.f90:341-374
enum { kMP_THOMPSON = 1, kMP_SB04 = 2, kMP_WSM6 = 4, kMP_WSM3 = 6 };

static bool cfg_ok(icar_hip_ctx *c, const char *who)
{
    if (c->step.failed) {
        icar_set_error(std::string(who) + ": this context abandoned a half-applied sub-step after an earlier error (the microphysics of that sub-step was "
                       "already applied when update_dt failed); its fields are not a model state -- reload them and set the clock (icar_hip_model_time_set)");
        return false;
    }
    if (c->step.configured) return true;
    icar_set_error(std::string(who) + ": call icar_hip_step_configure first");
    return false;
}

// ---- M0: mp(domain, options, dt, halo, subset) (mp_driver.f90:673-772) ------------------------------------------------
static int process_subdomain(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte)
{
    const icar_hip_step_config &g = c->step.cfg;
    if (ite < its || jte < jts) return 0;
    switch (g.microphysics) {
    case kMP_SB04:     return icar_mp_simple_run(c, dt, its, ite, jts, jte, kts, kte, nullptr);
    case kMP_WSM6:     return icar_wsm6_run(c, dt, its, ite, jts, jte, kts, kte);
    case kMP_WSM3:     return icar_wsm3_run(c, dt, its, ite, jts, jte, kts, kte);
    case kMP_THOMPSON: return icar_thompson_run(c, dt, its, ite, jts, jte, kts, kte, g.ids, g.ide, g.jds, g.jde, g.kds, g.kde);
    }
    icar_set_error("mp: microphysics option not built (1 Thompson, 2 mp_simple, 4 WSM6, 6 WSM3)");
    return 1;
}

int icar_mp_run(icar_hip_ctx *c, double dt_in, int halo, int subset)
{
    const icar_hip_step_config &g = c->step.cfg;
    if (g.microphysics == 0) return 0;
    IcarStepState &st = c->step;
    // the reference passes real(dt%seconds()) -- a REAL(4) -- into this arithmetic (time_step.f90:512, mp_driver.f90:673)
    const double dt4 = (double)(float)dt_in;
    const double upd = (double)g.mp_update_interval, now = st.model_time;
    if (st.mp_last_model_time == -999.0) st.mp_last_model_time = now - (upd > dt4 ? upd : dt4);           // :698-702
    if (((now + dt4) - st.mp_last_model_time) < upd) return 0;                                            // :705
    const float mp_dt = (float)(now - st.mp_last_model_time);                                             // :708
    if (halo < 0) st.mp_last_model_time = now;                                                             // :711-713 (not on the halo pass)
    int kte = g.kte;
    if (g.top_mp_level > 0 && g.top_mp_level < kte) kte = g.top_mp_level;                                  // :716-718
    int t[4][4];
    if (subset >= 0) {                                                                                     // :728-737
        icar_hip_mp_tiles(g.its, g.ite, g.jts, g.jte, 0, subset, t);
        if (process_subdomain(c, mp_dt, t[0][0], t[0][1], t[0][2], t[0][3], g.kts, kte)) return 1;
    }
    if (halo >= 0) {                                                                                       // :721-726 -> process_halo :609-658
        const int n = icar_hip_mp_tiles(g.its, g.ite, g.jts, g.jte, halo, 0, t);
        // a tile narrower than 2*halo makes opposite strips overlap: the reference then runs those columns once per strip, one
        // strip after the other -- a single batched launch would race on them
        const bool overlapping = (g.ite - g.its + 1 < 2 * halo) || (g.jte - g.jts + 1 < 2 * halo);
        int live[4][4], nl = 0;
        for (int s = 0; s < n; ++s) if (t[s][1] >= t[s][0] && t[s][3] >= t[s][2]) { memcpy(live[nl], t[s], sizeof live[nl]); ++nl; }
        const bool batched = !overlapping && (g.microphysics == kMP_THOMPSON || g.microphysics == kMP_SB04 || g.microphysics == kMP_WSM6);
        if (batched && nl) {
            int r = 0;
            if (g.microphysics == kMP_THOMPSON) r = icar_thompson_run_tiles(c, mp_dt, nl, live, g.kts, kte, g.ids, g.ide, g.jds, g.jde, g.kds, g.kde);
            else if (g.microphysics == kMP_WSM6) r = icar_wsm6_run_tiles(c, mp_dt, nl, live, g.kts, kte);
            else r = icar_mp_simple_run_tiles(c, mp_dt, nl, live, g.kts, kte, nullptr);
            if (r) return 1;
        } else if (!batched) {
            for (int s = 0; s < n; ++s) if (process_subdomain(c, mp_dt, t[s][0], t[s][1], t[s][2], t[s][3], g.kts, kte)) return 1;
        }
    }
    if (halo < 0 && subset < 0)
        if (process_subdomain(c, mp_dt, g.its, g.ite, g.jts, g.jte, g.kts, kte)) return 1;                 // :739-741
    return 0;
}

// ---- A1 bookkeeping: the Courant winds belong to (scheme, dt, advect_density) and to the wind state they were made from --
static bool winds_prepared(icar_hip_ctx *c, float dt)
{
    const icar_hip_step_config &g = c->step.cfg;
    return c->winds_valid && c->step.winds_scheme == g.advection && c->step.winds_dt == dt && c->step.winds_dens == (g.advect_density ? 1 : 0);
}

// with_wreal: w_real of diagnostic_update from the same winds in the same launch (advect.hip); *wreal_done: it was written
static int setup_winds(icar_hip_ctx *c, float dt, bool with_wreal = false, bool *wreal_done = nullptr)
{
    const icar_hip_step_config &g = c->step.cfg;
    return icar_advect_setup_winds(c, g.advection, dt, g.dx, g.advect_density, with_wreal, wreal_done);   // records (scheme, dt, density) in c->step
}

int icar_step_advect(icar_hip_ctx *c, float dt)
{   // advection_driver.f90:51-77
    const icar_hip_step_config &g = c->step.cfg;
    if (g.advection != ICAR_ADV_UPWIND && g.advection != ICAR_ADV_MPDATA) return 0;
    if (!winds_prepared(c, dt) && setup_winds(c, dt)) return 1;
    return icar_advect_run(c, g.advection, g.mpdata_order, g.flux_corrected_transport, g.advect_density, g.advect_fields, g.n_advect);
}

// ---- T2: compute_dt + update_dt (time_step.f90:217-330, :375-423) ------------------------------------------------------
static int compute_dt(icar_hip_ctx *c, float *dt_out, bool *on_device)
{
    const icar_hip_step_config &g = c->step.cfg;
    const float *dzl = c->step.dz_levels.data();
    const int strict = g.cfl_strictness;
    float mu = 0, mv = 0, mw = 0, maxwind1d = 0, maxwind3d = 0;
    *on_device = false;
    if (strict == 1 || strict == 2 || strict == 5) {
        float m3[3];
        if (icar_max_abs_winds_run(c, m3)) return 1;
        mu = m3[0]; mv = m3[1]; mw = m3[2];
    }
    const float sqrt3 = sqrtf(3.0f) * 1.001f;                                     // :229
    if (strict == 1) { maxwind1d = fmaxf(fmaxf(mu, mv), mw); maxwind3d = maxwind1d * sqrt3; }              // :238-246
    else if (strict == 5) maxwind3d = (mu + mv) + mw;                                                          // :248-259
    else {
        // strictness 2, 3, 4: the per-cell Courant sum (:264-289).  With RCCL and 3 / 4 the tile maximum stays on the device
        // until it has been reduced over the images (dt = factor / max is monotone: min dt == factor / max, bit for bit)
        float *d_val = c->d_red + 12;
        if ((strict == 3 || strict == 4) && c->comm && icar_hip_comm_kind(c) == ICAR_COMM_RCCL) {
            // the last sub-step has usually left the GLOBAL maximum behind (reduced over the images beside its advection)
            if (icar_cfl_prefetched_global(c, g.dx, dzl, &maxwind3d)) { *on_device = true; goto have_max; }
            if (icar_max_courant_run(c, g.dx, dzl, nullptr, d_val)) return 1;
            if (icar_comm_max_device(c, d_val) != 0) return 1;
            if (!c->step.h_val) HIPCHK(hipHostMalloc((void **)&c->step.h_val, sizeof(float), hipHostMallocDefault));
            HIPCHK(hipMemcpyAsync(c->step.h_val, d_val, sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            maxwind3d = *c->step.h_val;
            *on_device = true;
        } else if (icar_max_courant_run(c, g.dx, dzl, &maxwind3d, nullptr)) return 1;
    have_max:
        if (strict == 2) {                                                                                     // :291-300
            maxwind3d = maxwind3d * 0.577350269f;
            maxwind1d = fmaxf(fmaxf(mu, mv), mw);
            maxwind3d = fmaxf(maxwind1d, maxwind3d);
        } else if (strict == 4) maxwind3d = maxwind3d * sqrt3;                                                  // :302-305
    }
    if (!(maxwind3d == maxwind3d)) {
        if (c->cfl_wait_failed) { c->cfl_wait_failed = false; icar_set_error("compute_dt: the prefetched CFL maximum could not be read (HIP event wait failed)"); }
        else icar_set_error("compute_dt: the CFL maximum is NaN (the winds contain NaN)");
        return 1;
    }
    const float dt = g.cfl_reduction_factor / maxwind3d;                                                       // :313
    if (dt < 1e-1f) { icar_set_error("ERROR time step too small"); return 1; }                                 // :322-328 `stop`
    *dt_out = dt;
    return 0;
}

int icar_update_dt(icar_hip_ctx *c, double *seconds)
{
    float dt; bool reduced;
    if (compute_dt(c, &dt, &reduced)) return 1;
    double s = (double)dt;
    if (!reduced && icar_comm_co_reduce(c, &s, true)) return 1;                   // :413 co_min(seconds)
    *seconds = s < 120.0 ? s : 120.0;                                             // :417
    return 0;
}

// ---- T1: one pass of time_step.f90:474-539 ----------------------------------------------------------------------------
static int halo_send(icar_hip_ctx *c) { const icar_hip_step_config &g = c->step.cfg; return icar_comm_halo_send(c, g.halo_size, g.exchange_fields, g.n_exchange); }
static int halo_retrieve(icar_hip_ctx *c) { const icar_hip_step_config &g = c->step.cfg; return icar_comm_halo_retrieve(c, g.halo_size, g.exchange_fields, g.n_exchange); }

struct AuxScope {          // entry points called while this object lives launch on the second stream; leaves it on every return path
    icar_hip_ctx *c; bool on = false;
    explicit AuxScope(icar_hip_ctx *c_) : c(c_) {}
    int begin() { if (icar_hip_aux_begin(c)) return 1; on = true; return 0; }
    void end() { if (on) { icar_hip_aux_end(c); on = false; } }
    ~AuxScope() { end(); }
};

// The opening of a sub-step does not depend on its dt: diagnostic_update, the interior microphysics (its own time step is
// model_time - last_model_time, mp_driver.f90:708) and, on the second stream, the strips, the pack + halo transfer and the
// interface diagnostics.  The step loops issue it BEFORE they wait for the CFL maximum of update_dt -- which the previous
// sub-step left reducing beside its advection -- so that the device has ~the whole microphysics queued while the host reads dt
// (otherwise every step starts with the host's wake-up + launch latency, ~20 us: 4 % of the 8-GPU tile's step).  Same launches,
// same operands, same results; only the order of issue changes.  Possible when nothing of the opening needs dt: no
// mp_update_interval gating, not the first microphysics call, halo_size 1, a prefetched CFL maximum waiting (update_dt will not
// launch anything on the main stream) and no end-time clamp in reach (the caller checks the last one).
bool icar_substep_can_open_early(icar_hip_ctx *c)
{
    const icar_hip_step_config &g = c->step.cfg;
    return g.microphysics != 0 && g.halo_size == 1 && g.mp_update_interval == 0.0f && c->step.mp_last_model_time != -999.0
        && g.prefetch_dt && (g.cfl_strictness == 3 || g.cfl_strictness == 4) && icar_cfl_prefetch_waiting(c) && !c->on_aux;
}

static int substep_open(icar_hip_ctx *c, double dt, bool dt_known, bool &wreal_later, bool &face_later)
{
    const icar_hip_step_config &g = c->step.cfg;
    const bool adv = (g.advection == ICAR_ADV_UPWIND || g.advection == ICAR_ADV_MPDATA);
    wreal_later = false; face_later = false;
    if (g.diagnostics) {
        if (g.microphysics != kMP_WSM3) {
            face_later = true;
            if (icar_diagnostic_update_run(c, ICAR_DIAG_CELL)) return 1;
            wreal_later = true;
        } else if (icar_diagnostic_update_run(c, 3)) return 1;
    }
    if (icar_hip_aux_fork(c)) return 1;
    const double mp_last_before = c->step.mp_last_model_time;
    if (icar_mp_run(c, dt, -1, 1)) return 1;                                      // :523 interior
    const double mp_last_after = c->step.mp_last_model_time;
    c->step.mp_last_model_time = mp_last_before;
    {
        AuxScope aux(c);
        if (aux.begin()) return 1;
        if (icar_mp_run(c, dt, 1, -1)) return 1;                                  // :512 strips
        if (halo_send(c)) return 1;                                               // :515 pack + RCCL send / recv
        if (halo_retrieve(c)) return 1;                                           // :526 unpack, see icar_substep
        if (face_later && icar_diagnostic_update_run(c, ICAR_DIAG_FACE)) return 1;
        if (dt_known && adv && setup_winds(c, (float)dt)) return 1;
    }
    c->step.mp_last_model_time = mp_last_after;
    return 0;
}

int icar_substep_open_early(icar_hip_ctx *c)
{
    bool wl, fl;
    if (substep_open(c, 0.0, false, wl, fl)) return 1;
    c->step.early_open = true; c->step.early_wreal = wl; c->step.early_face = fl;
    return 0;
}

// update_dt failed after the opening of the sub-step was issued (time step too small, a transport timeout): join the second
// stream and forget the opening, so that the next call on this context does not continue a half-issued sub-step.  The opening has
// already applied the interior and strip microphysics of the abandoned sub-step and advanced mp_last_model_time: a caller that
// retried would run the microphysics twice for one model time.  The reference STOPs where update_dt fails (time_step.f90:322-328);
// here the context is marked failed and every later stepping call returns an error until the caller reloads the fields and
// sets the model clock (icar_hip_model_time_set clears the mark: that is what a restart does).
static int update_dt_opened(icar_hip_ctx *c, double *dt)
{
    if (icar_update_dt(c, dt) == 0) return 0;
    if (c->step.early_open) {
        const std::string why = icar_hip_last_error();                   // (aux_join may overwrite it)
        (void)icar_hip_aux_join(c); c->step.early_open = false; c->step.failed = true;
        icar_set_error(why + " [the sub-step was already opened: context marked failed]");
    }
    return 1;
}

int icar_substep(icar_hip_ctx *c, double dt, bool enforce)
{
    const icar_hip_step_config &g = c->step.cfg;
    const float dtf = (float)dt;
    const bool adv = (g.advection == ICAR_ADV_UPWIND || g.advection == ICAR_ADV_MPDATA);
    const bool stepping = dt > 1e-3;                                              // :483
    bool wreal_later = false, face_later = false, wreal_done = false;
    const bool early = c->step.early_open;
    c->step.early_open = false;
    if (early) {
        // the opening is in flight (icar_substep_open_early); what it left out: the wind setup of the advect() below, on the second stream
        wreal_later = c->step.early_wreal; face_later = c->step.early_face;
        if (!stepping) { icar_set_error("substep: opened early but dt <= 1e-3"); return 1; }
        if (adv) {
            AuxScope aux(c);
            if (aux.begin()) return 1;
            if (setup_winds(c, dtf, wreal_later, &wreal_done)) return 1;
        }
        if (icar_hip_aux_join(c)) return 1;
    } else
    if (g.diagnostics) {                                                          // :474
        if (g.microphysics != kMP_WSM3) {                                         // WSM3 reads w_real
            // exner / T / density now; the interface values and mass-point winds (nothing the microphysics reads or writes)
            // beside the interior launch below
            face_later = stepping && g.microphysics != 0;
            if (icar_diagnostic_update_run(c, face_later ? ICAR_DIAG_CELL : 1)) return 1;
            if (stepping) wreal_later = true;                                     // beside the advection, below
            else if (icar_diagnostic_update_run(c, 2)) return 1;
        } else if (icar_diagnostic_update_run(c, 3)) return 1;
    }
    if (!stepping) return 0;

    // :512-526  mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve
    if (early) {
    } else if (g.microphysics != 0 && g.halo_size > 1) {
        // halo_send packs the first halo_size owned rows / columns, mp(halo=1) only processes the outermost one: with a halo wider
        // than 1 the message carries cells the interior pass has not touched yet (the reference sends them in that state,
        // time_step.f90:512-523).  Beside the interior launch the pack would read them while they are written, so this
        // configuration keeps the reference's order on one stream.
        if (icar_mp_run(c, dt, 1, -1)) return 1;                                  // :512
        if (halo_send(c)) return 1;                                               // :515
        if (icar_mp_run(c, dt, -1, 1)) return 1;                                  // :523
        if (face_later && icar_diagnostic_update_run(c, ICAR_DIAG_FACE)) return 1;
        if (halo_retrieve(c)) return 1;                                           // :526
    } else if (g.microphysics != 0) {
        // The interior launch is the critical path of this block, so IT stays on the main stream (diag -> interior -> unpack ->
        // advect are then same-stream neighbours); the side work -- strips, pack + transfer, and the wind setup of the advect()
        // that follows -- goes to the second stream and has finished long before the join.  (Rounds 1-2 had it the other way
        // round: every edge of the critical path was then a cross-stream event, ~20 us each on this runtime.)
        if (icar_hip_aux_fork(c)) return 1;
        // The interior launch is ISSUED first: on a small tile the host's launch rate is the limit, and the ~8 launches of the side
        // work in front of it delayed the critical kernel by ~0.1 ms.  mp(halo=1) must still see the clock state it sees in the
        // reference's order (it runs before the interior pass moves last_model_time, :711-713), so that state is put back for it.
        const double mp_last_before = c->step.mp_last_model_time;
        if (icar_mp_run(c, dt, -1, 1)) return 1;                                  // :523 interior
        const double mp_last_after = c->step.mp_last_model_time;
        c->step.mp_last_model_time = mp_last_before;
        {
            AuxScope aux(c);
            if (aux.begin()) return 1;
            if (icar_mp_run(c, dt, 1, -1)) return 1;                              // :512 strips (the halo pass leaves last_model_time alone, :711)
            if (halo_send(c)) return 1;                                           // :515 pack + RCCL send / recv
            // :526 halo_retrieve fills the cells OUTSIDE the owned tile; the microphysics is column-local and runs on owned
            // columns only, so the interior pass neither reads nor writes them: the unpack follows the receive on the second
            // stream instead of waiting for the interior launch on the main one (one kernel + one dependency edge off the
            // critical path: ~15 us per sub-step)
            if (halo_retrieve(c)) return 1;
            if (face_later && icar_diagnostic_update_run(c, ICAR_DIAG_FACE)) return 1;
            // the Courant winds (and MPDATA coefficients) read u, v, w, density and the jacobians, none of which the microphysics
            // touches: streaming kernels beside the interior launch.  w_real of diagnostic_update (from these winds, before
            // their forcing) comes out of the same launch: beside the advection its reads cost the MPDATA kernel 40 us
            if (adv && setup_winds(c, dtf, wreal_later, &wreal_done)) return 1;
        }
        c->step.mp_last_model_time = mp_last_after;
        if (icar_hip_aux_join(c)) return 1;
    } else {
        if (halo_send(c)) return 1;
        if (halo_retrieve(c)) return 1;
    }

    // :529-534  advect, with the whole-field forcing that does not depend on it (and the next CFL reduction) beside it.
    // (Measured alternative: w_real, the forcing of u, v, w and the CFL reduction beside the interior microphysics instead.  The
    // advection alone takes 1.02 ms instead of 1.12 with them in its shadow, but its persistent blocks leave them unused register
    // slots, while beside Thompson every streaming wave displaces a compute wave: 3.20 instead of 3.08 ms per step.)
    int aside_f[16], aside_b[16], na = 0, rest_f[16], rest_b[16], nr = 0;
    for (int m = 0; m < g.n_forced; ++m) {
        const int f = g.forced_fields[m];
        const bool whole = !g.force_boundaries[m] && (f == ICAR_F_U || f == ICAR_F_V || f == ICAR_F_W || f == ICAR_F_PRESSURE);
        if (whole) { aside_f[na] = f; aside_b[na] = 0; ++na; } else { rest_f[nr] = f; rest_b[nr] = g.force_boundaries[m]; ++nr; }
    }
    const bool cfl_ahead = g.prefetch_dt && (g.cfl_strictness == 3 || g.cfl_strictness == 4);
    const bool beside = na > 0 || cfl_ahead || wreal_later;
    // the second stream starts from the state BEFORE the advection -- but after the wind setup, which reads the winds the
    // forcing beside the advection rewrites
    if (adv && !winds_prepared(c, dtf) && setup_winds(c, dtf)) return 1;
    if (beside && icar_hip_aux_fork(c)) return 1;
    if (icar_step_advect(c, dtf)) return 1;                                       // :529
    if (beside) {
        {
            AuxScope aux(c);
            if (aux.begin()) return 1;
            if (wreal_later && !wreal_done && icar_diagnostic_update_run(c, 2)) return 1;   // :165-194, from the winds of this step (before their forcing)
            if (na && icar_apply_forcing_run(c, dt, aside_f, aside_b, na, g.west_boundary, g.east_boundary, g.south_boundary, g.north_boundary)) return 1;
            // (with RCCL the tile maximum is also all-reduced over the images here, in the advection's shadow)
            if (cfl_ahead && icar_max_courant_prefetch_run(c, g.dx, c->step.dz_levels.data(), c->comm && icar_hip_comm_kind(c) == ICAR_COMM_RCCL)) return 1;
        }
    }
    // the side work touches u, v, w, p, w_real and the CFL scalar; what is left on the main stream touches advected scalars only, so
    // the join can wait until after it (on a small tile the side chain is as long as the advection: its tail then overlaps the
    // boundary-ring forcing instead of delaying it).  Any other forced field keeps the join in front.
    bool scalars_only = true;
    for (int m = 0; m < nr; ++m) scalars_only = scalars_only && rest_f[m] >= 0 && rest_f[m] < ICAR_N_ADVECTABLE;
    if (beside && !scalars_only && icar_hip_aux_join(c)) return 1;
    if (nr && icar_apply_forcing_run(c, dt, rest_f, rest_b, nr, g.west_boundary, g.east_boundary, g.south_boundary, g.north_boundary)) return 1;   // :534
    if (enforce && icar_enforce_limits_run(c, g.advect_fields, g.n_advect)) return 1;                                                                // :537-539
    if (beside && scalars_only && icar_hip_aux_join(c)) return 1;
    return 0;
}

// ---- update_winds (wind.f90:289-369) and iterative_winds (:371-498) -----------------------------------------------------
// windtype: src/constants/icar_constants.f90:368-377
enum { kWIND_LINEAR = 1, kCONSERVE_MASS = 2, kITERATIVE_WINDS = 3, kLINEAR_ITERATIVE_WINDS = 5 };

static int iterative_winds(icar_hip_ctx *c, float dx, int wind_iterations, int halo, int update)
{
    // :404-405 exchange_u / exchange_v, :407-415 balance_uvw, :430-441 the model-top w spread over the column, then
    // `do it = 0, wind_iterations`: one sweep (:455-481) and the exchanges (:482-483)
    if (icar_comm_exchange_uv(c, halo, update)) return 1;
    if (icar_balance_uvw_run(c, dx, update)) return 1;
    if (icar_iterative_winds_correct_w(c, update)) return 1;
    const int n = wind_iterations + 1;
    if (!icar_comm_has_peers(c)) return icar_iterative_winds_sweep(c, dx, n, update);        // no neighbour to talk to: the whole loop in one call
    for (int it = 0; it < n; ++it) {
        if (icar_iterative_winds_sweep(c, dx, 1, update)) return 1;
        if (icar_comm_exchange_uv(c, halo, update)) return 1;
    }
    return 0;
}

int icar_update_winds(icar_hip_ctx *c, int windtype, int wind_iterations, float dx, int halo, int update)
{
    if (windtype != 0 && windtype != kWIND_LINEAR && windtype != kCONSERVE_MASS && windtype != kITERATIVE_WINDS && windtype != kLINEAR_ITERATIVE_WINDS) {
        icar_set_error("update_winds: windtype is 0, 1 (linear), 2 (conserve mass), 3 (iterative) or 5 (linear + iterative)"); return 1;
    }
    if (icar_make_winds_grid_relative(c, update)) return 1;                                              // :300 / :338
    if ((windtype == kWIND_LINEAR || windtype == kLINEAR_ITERATIVE_WINDS) && icar_spatial_winds_run(c, update)) return 1;   // linear_perturb
    if (windtype == kCONSERVE_MASS && icar_mass_conservative_acceleration(c, update)) return 1;
    if ((windtype == kITERATIVE_WINDS || windtype == kLINEAR_ITERATIVE_WINDS) && iterative_winds(c, dx, wind_iterations, halo, update)) return 1;
    return icar_balance_uvw_run(c, dx, update);                                                          // :329-331 / :357-363
}

extern "C" {

// update_winds(domain, options): the first call works on u, v, w (update = 0), every later one on their dqdt_3d (the next
// forcing step's winds); update < 0 = as the reference decides it (first call of this context or not, wind.f90:297).
int icar_hip_update_winds(icar_hip_ctx *c, int windtype, int wind_iterations, float dx, int halo, int update)
{
    if (!c) { icar_set_error("update_winds: null ctx"); return 1; }
    if (wind_iterations < 0 || halo < 1) { icar_set_error("update_winds: wind_iterations >= 0, halo >= 1"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    const int upd = update < 0 ? (c->step.winds_first ? 0 : 1) : (update ? 1 : 0);
    if (icar_update_winds(c, windtype, wind_iterations, dx, halo, upd)) return 1;
    c->step.winds_first = false;
    if (!upd) icar_winds_changed(c);
    return 0;
}

int icar_hip_step_configure(icar_hip_ctx *c, const icar_hip_step_config *cfg, const float *dz_levels)
{
    if (!c || !cfg || !dz_levels) { icar_set_error("step_configure: null argument"); return 1; }
    if (cfg->n_advect < 0 || cfg->n_advect > ICAR_N_ADVECTABLE || cfg->n_exchange < 0 || cfg->n_exchange > ICAR_N_ADVECTABLE ||
        cfg->n_forced < 0 || cfg->n_forced > 16) { icar_set_error("step_configure: bad list length"); return 1; }
    for (int m = 0; m < cfg->n_advect; ++m) if (cfg->advect_fields[m] < 0 || cfg->advect_fields[m] >= ICAR_N_ADVECTABLE) { icar_set_error("step_configure: advect_fields holds advectable scalars"); return 1; }
    for (int m = 0; m < cfg->n_exchange; ++m) if (cfg->exchange_fields[m] < 0 || cfg->exchange_fields[m] >= ICAR_N_ADVECTABLE) { icar_set_error("step_configure: exchange_fields holds advectable scalars"); return 1; }
    if (cfg->cfl_strictness < 1 || cfg->cfl_strictness > 5) { icar_set_error("step_configure: cfl_strictness is 1..5"); return 1; }
    if (cfg->halo_size < 1) { icar_set_error("step_configure: halo_size >= 1"); return 1; }
    if (c->d.nz > 4096) { icar_set_error("step_configure: the CFL reduction holds dz_levels of at most 4096 levels"); return 1; }
    if (cfg->its < c->ims || cfg->ite > c->ime || cfg->jts < c->jms || cfg->jte > c->jme || cfg->kts < c->kms || cfg->kte > c->kme) {
        icar_set_error("step_configure: its..kte outside the ims..kme of the context"); return 1;
    }
    c->step.cfg = *cfg;
    c->step.dz_levels.assign(dz_levels, dz_levels + c->d.nz);
    c->step.configured = true;
    return 0;
}

int icar_hip_model_time_set(icar_hip_ctx *c, double seconds) { if (!c) { icar_set_error("null ctx"); return 1; } c->step.model_time = seconds; c->step.failed = false; return 0; }
double icar_hip_model_time(const icar_hip_ctx *c) { return c ? c->step.model_time : 0.0; }
int icar_hip_mp_reset(icar_hip_ctx *c) { if (!c) { icar_set_error("null ctx"); return 1; } c->step.mp_last_model_time = -999.0; return 0; }

int icar_hip_mp(icar_hip_ctx *c, double dt, int halo, int subset)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!cfg_ok(c, "mp")) return 1;
    HIPCHK(hipSetDevice(c->device));
    return icar_mp_run(c, dt, halo, subset);
}

int icar_hip_advect_step(icar_hip_ctx *c, double dt)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!cfg_ok(c, "advect_step")) return 1;
    HIPCHK(hipSetDevice(c->device));
    return icar_step_advect(c, (float)dt);
}

int icar_hip_compute_dt(icar_hip_ctx *c, double *dt_seconds)
{
    if (!c || !dt_seconds) { icar_set_error("compute_dt: null argument"); return 1; }
    if (!cfg_ok(c, "compute_dt")) return 1;
    HIPCHK(hipSetDevice(c->device));
    float dt; bool reduced;
    IcarComm *keep = c->comm; c->comm = nullptr;          // this image alone: no reduction over the images
    const int r = compute_dt(c, &dt, &reduced);
    c->comm = keep;
    if (r) return 1;
    *dt_seconds = (double)dt;
    return 0;
}

int icar_hip_update_dt(icar_hip_ctx *c, double *dt_seconds)
{
    if (!c || !dt_seconds) { icar_set_error("update_dt: null argument"); return 1; }
    if (!cfg_ok(c, "update_dt")) return 1;
    HIPCHK(hipSetDevice(c->device));
    return icar_update_dt(c, dt_seconds);
}

int icar_hip_substep(icar_hip_ctx *c, double dt_seconds, int enforce_limits)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!cfg_ok(c, "substep")) return 1;
    if (c->on_aux) { icar_set_error("substep: called between aux_begin and aux_end"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_substep(c, dt_seconds, enforce_limits != 0);
}

// nsteps sub-steps of the loop without an end time (a benchmark's "K passes of the hot path", a host that counts steps):
// update_dt -> substep -> clock += dt, nothing of the host in between.  dt_last receives the last step's dt.
int icar_hip_step_n(icar_hip_ctx *c, int nsteps, double *dt_last)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!cfg_ok(c, "step_n")) return 1;
    if (c->on_aux) { icar_set_error("step_n: called between aux_begin and aux_end"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    double dt = 0.0;
    for (int n = 0; n < nsteps; ++n) {
        if (icar_substep_can_open_early(c) && icar_substep_open_early(c)) return 1;
        if (update_dt_opened(c, &dt)) return 1;
        if (icar_substep(c, dt, false)) return 1;
        c->step.model_time += dt;
    }
    if (dt_last) *dt_last = dt;
    return 0;
}

int icar_hip_step(icar_hip_ctx *c, double end_time_seconds, int *nsteps)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!cfg_ok(c, "step")) return 1;
    if (c->on_aux) { icar_set_error("step: called between aux_begin and aux_end"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    int n = 0;
    while (c->step.model_time < end_time_seconds) {                              // :462
        double dt;
        // (update_dt caps dt at 120 s, :417: farther than that from the end no clamp can shorten this step)
        if (end_time_seconds - c->step.model_time > 120.0 && icar_substep_can_open_early(c) && icar_substep_open_early(c)) return 1;
        if (update_dt_opened(c, &dt)) return 1;                                  // :465
        if (c->step.model_time + dt > end_time_seconds) dt = end_time_seconds - c->step.model_time;       // :469-471
        if (icar_substep(c, dt, (end_time_seconds - c->step.model_time) < dt * 2)) return 1;               // :474-539
        c->step.model_time += dt;                                                // :547
        ++n;
    }
    if (nsteps) *nsteps = n;
    return 0;
}

}  // extern "C"
