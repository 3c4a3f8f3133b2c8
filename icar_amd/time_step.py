"""time stepper mirror (src/main/time_step.f90): compute_dt / update_dt / step."""
import ctypes
import numpy as np
from .capi import lib, check, IcarHipError
from .advection import advect
from .microphysics import mp
from .halo import co_min


def compute_dt(domain, options):
    """time_step.f90:217-330 for every cfl_strictness (1..5; 3 is the default): the reductions run on the device, the few
    REAL(4) operations that combine them are the reference's, in its order (units included: settings 1 and 5 compare
    m/s with a Courant number, as the reference does).  use_density is not on the path (its branches are empty)."""
    f32 = np.float32
    strict = int(options.parameters.cfl_strictness)
    maxwind1d = maxwind3d = f32(0)
    if strict in (1, 2, 5):
        m = (ctypes.c_float * 3)()
        check(lib().icar_hip_max_abs_winds(domain.ctx, m), "icar_hip_max_abs_winds")
        mu, mv, mw = f32(m[0]), f32(m[1]), f32(m[2])
    sqrt3 = f32(f32(np.sqrt(f32(3.0))) * f32(1.001))
    if strict == 1:
        maxwind1d = max(max(mu, mv), mw)
        maxwind3d = f32(maxwind1d * sqrt3)
    elif strict == 5:
        maxwind3d = f32(f32(mu + mv) + mw)
    else:
        dzl = np.ascontiguousarray(options.parameters.dz_levels, np.float32)
        out = ctypes.c_float()
        check(lib().icar_hip_max_courant(domain.ctx, ctypes.c_float(domain.dx), dzl.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.byref(out)), "icar_hip_max_courant")
        maxwind3d = f32(out.value)
        if strict == 2:
            maxwind3d = f32(maxwind3d * f32(0.577350269))
            maxwind1d = max(max(mu, mv), mw)
            maxwind3d = max(maxwind1d, maxwind3d)
        elif strict == 4:
            maxwind3d = f32(maxwind3d * sqrt3)
    dt = f32(options.parameters.cfl_reduction_factor) / f32(maxwind3d)
    if dt < 1e-1:
        raise IcarHipError("ERROR time step too small")      # time_step.f90:322-328 `stop`
    return float(dt)


def update_dt(domain, options, group=None, device=None):
    """time_step.f90:375-423: local CFL dt, co_min over images, cap at 120 s.

    With RCCL (several images, cfl_strictness 3 or 4) the reduction never leaves the device before the all-reduce:
    k_max_courant writes the tile's maximum Courant sum into a 1-element device tensor, all_reduce(MAX) runs on it, and
    one read brings the global value back -- dt = factor / max is monotone, so min over images of dt == factor / max
    over images, bit for bit.  Other settings / backends combine on the host (compute_dt) and co_min the REAL(8)."""
    strict = int(options.parameters.cfl_strictness)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and strict in (3, 4):
        import torch
        if dist.get_backend(group) == "nccl":                 # also with one rank: the same RCCL call as with eight
            f32 = np.float32
            t = getattr(domain, "_cfl_dev", None)
            if t is None:
                t = domain._cfl_dev = torch.zeros(1, dtype=torch.float32, device=f"cuda:{domain.device}")
            dzl = np.ascontiguousarray(options.parameters.dz_levels, np.float32)
            check(lib().icar_hip_max_courant_device(domain.ctx, ctypes.c_float(domain.dx), dzl.ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.c_void_p(t.data_ptr())), "icar_hip_max_courant_device")
            if domain.needs_host_sync():
                domain.synchronize()                   # the reduction ran on a stream RCCL does not order itself against
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            maxwind3d = f32(t.item())
            if strict == 4:
                maxwind3d = f32(maxwind3d * f32(f32(np.sqrt(f32(3.0))) * f32(1.001)))
            dt = f32(options.parameters.cfl_reduction_factor) / f32(maxwind3d)
            if dt < 1e-1:
                raise IcarHipError("ERROR time step too small")
            return min(float(dt), 120.0)
    seconds = co_min(compute_dt(domain, options), group=group, device=device)
    return min(seconds, 120.0)


def mp_and_halo(domain, options, dt, overlap=True, prepare_advection=True, beside_interior=()):
    """time_step.f90:512-526: mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve.

    The strips, the pack kernels and the exchange stay on the context's main stream; the interior launch runs beside them
    on the context's second stream -- a one-cell-wide strip launch cannot fill 256 CUs, and the interior
    does not have to wait for it (disjoint columns).  An image without neighbours runs exactly the same launches (its
    halo_send / halo_retrieve have nobody to talk to), so 1-image and N-image timings compare like with like.
    prepare_advection: also launch the wind setup of the advect() that follows, on the main stream beside the interior.
    beside_interior: further callables whose launches neither read what the microphysics writes nor write what it reads
    (the w_real part of diagnostic_update); they go out on the main stream beside the interior too, or right away without overlap."""
    from .constants import kADV_UPWIND, kADV_MPDATA
    overlap = overlap and options.physics.microphysics != 0    # every scheme's calls on disjoint tiles share no scratch
    if overlap:
        domain.aux_fork()
    mp(domain, options, dt, halo=1)                            # :512
    domain.halo_send()                                         # :515
    if overlap:
        domain.aux_begin()
        try:
            mp(domain, options, dt, subset=1)                  # :523
        finally:
            domain.aux_end()
        if prepare_advection and options.physics.advection in (kADV_UPWIND, kADV_MPDATA):
            # the Courant winds of the advect() that follows (setup_module_winds) read u, v, w, density and the jacobians, none
            # of which the microphysics touches: a streaming kernel on the main stream beside the VALU-bound interior launch
            from .advection import setup_winds
            setup_winds(domain, options, dt)
        for fn in beside_interior:
            fn()
        domain.aux_join()
    else:
        for fn in beside_interior:
            fn()
        mp(domain, options, dt, subset=1)
    domain.halo_retrieve()                                     # :526


# whole-field forcing of these members (domain_obj.f90:2427, :2436: `x += dqdt * dt`) touches nothing the advection reads
# (it works from the Courant winds of setup_module_winds, the scalars, density and the jacobians)
_FORCING_BESIDE_ADVECT = ("u", "v", "w", "pressure")


def substep(domain, options, dt, forced=None, diagnostics=True, enforce=False, prefetch_dt=True):
    """One pass of time_step.f90:474-539: diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve ->
    advect -> apply_forcing (-> enforce_limits), with the streaming kernels that do not depend on the two heavy ones issued
    beside them: the wind setup of advect() beside the interior microphysics (mp_and_halo); on the second stream beside the
    advection (which works from the Courant winds set up before, and whose blocks leave room for one small wave per SIMD) the
    w_real diagnostic (read by WSM3 and the output only: winds, slopes, jacobian -- nothing the two heavy kernels write), then
    the whole-field forcing of u, v, w, pressure, then the CFL reduction of the next update_dt (the library discards it if
    anything writes u, v, w before it is asked for).  Same launches, same operands, same results as the plain sequence."""
    from .constants import ADVECTION_ORDER, kMP_WSM3
    beside = ()
    wreal_later = False
    if diagnostics:
        if options.physics.microphysics != kMP_WSM3:           # WSM3 reads w_real
            domain.diagnostic_update(parts=1)                  # :474 (exner, density, ... before the microphysics)
            if dt > 1e-3:
                wreal_later = True                             # beside the advection, below
            else:
                beside = (lambda: domain.diagnostic_update(parts=2),)
        else:
            domain.diagnostic_update()
    if dt > 1e-3:                                              # :483
        mp_and_halo(domain, options, dt, beside_interior=beside)   # :512-526
        aside = [f for f in (forced or []) if not f[1] and f[0] in _FORCING_BESIDE_ADVECT]
        rest = [f for f in (forced or []) if f not in aside]
        cfl_ahead = prefetch_dt and int(options.parameters.cfl_strictness) in (3, 4)
        if aside or cfl_ahead or wreal_later:
            domain.aux_fork()                                  # the second stream starts from the state BEFORE the advection
        advect(domain, options, dt)                            # :529
        if aside or cfl_ahead or wreal_later:
            domain.aux_begin()
            try:
                if wreal_later:
                    domain.diagnostic_update(parts=2)          # :165-194, from the winds of this step (before their forcing)
                if aside:
                    domain.apply_forcing(dt, aside)            # :534, the part that does not wait for the advection
                if cfl_ahead:
                    domain.prefetch_courant(options)           # the reduction of the next update_dt (:217-330), winds now final
            finally:
                domain.aux_end()
            domain.aux_join()
        if rest:
            domain.apply_forcing(dt, rest)                     # :534, boundary relaxation of the advected scalars
        if enforce:                                            # :537-539
            names = [n for n in ADVECTION_ORDER if options.vars_to_advect.get(n, 0) > 0]
            domain.enforce_limits(names)
    else:
        for fn in beside:
            fn()


def step(domain, end_time, options, group=None, device=None, forced=None, diagnostics=True):
    """time_step.f90:440-551 for configurations 1-4 (rad/lsm/pbl/cu are no-ops there): the operator-split loop
         update_dt -> diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve
         -> advect -> apply_forcing -> enforce_limits (last two sub-steps).
    forced = [(member, force_boundaries), ...] with uploaded dqdt (domain.set_dqdt); None skips apply_forcing."""
    nsteps = 0
    while domain.model_time_seconds < end_time:
        dt = update_dt(domain, options, group=group, device=device)
        if domain.model_time_seconds + dt > end_time:          # :469-471
            dt = end_time - domain.model_time_seconds
        substep(domain, options, dt, forced=forced, diagnostics=diagnostics,
                enforce=(end_time - domain.model_time_seconds) < dt * 2)
        domain.model_time_seconds += dt                        # :547
        nsteps += 1
    return nsteps
