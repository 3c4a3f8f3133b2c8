"""time stepper mirror (src/main/time_step.f90): compute_dt / update_dt / step.

The loop itself lives behind the C ABI (icar_amd/csrc/timestep.hip: icar_hip_update_dt, icar_hip_substep, icar_hip_step), with
the two-stream choreography inside the library, so a Fortran host gets exactly what this module gets; the functions here
hand the library the options and call one entry point each."""
import ctypes
from .capi import lib, check


def compute_dt(domain, options):
    """time_step.f90:217-330 for every cfl_strictness (1..5; 3 is the default) on this image's tile: the reductions run on the
    device, the few REAL(4) operations that combine them are the reference's, in its order (units included: settings 1 and 5
    compare m/s with a Courant number, as the reference does).  use_density is not on the path (its branches are empty).
    Raises "ERROR time step too small" where the reference stops (:322-328)."""
    domain.configure(options)
    dt = ctypes.c_double()
    check(lib().icar_hip_compute_dt(domain.ctx, ctypes.byref(dt)), "icar_hip_compute_dt")
    return dt.value


def update_dt(domain, options):
    """time_step.f90:375-423: local CFL dt, co_min over the images of the domain's communicator, cap at 120 s.

    With RCCL and cfl_strictness 3 or 4 the reduction never leaves the device before the all-reduce: k_max_courant leaves the
    tile's maximum Courant sum in device memory, ncclAllReduce(MAX) runs on it, one read brings the global value back --
    dt = factor / max is monotone, so min over images of dt == factor / max over images, bit for bit.  Other settings /
    transports combine on the host and co_min the REAL(8)."""
    domain.configure(options)
    dt = ctypes.c_double()
    check(lib().icar_hip_update_dt(domain.ctx, ctypes.byref(dt)), "icar_hip_update_dt")
    return dt.value


def mp_and_halo(domain, options, dt):
    """time_step.f90:512-526 alone: mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve, the strips, the pack kernels and
    the exchange on the context's main stream, the interior launch beside them on its second stream -- icar_hip_substep with the
    rest of the sub-step (diagnostic_update, advect, apply_forcing) switched off."""
    keep = (domain._forced, domain._diagnostics, domain._prefetch_dt)
    domain.configure(options, forced=(), diagnostics=False, prefetch_dt=False, advection=0)
    try:
        check(lib().icar_hip_substep(domain.ctx, float(dt), 0), "icar_hip_substep")
    finally:
        domain._forced, domain._diagnostics, domain._prefetch_dt = keep


def substep(domain, options, dt, forced=None, diagnostics=True, enforce=False, prefetch_dt=True):
    """One pass of time_step.f90:474-539: diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve ->
    advect -> apply_forcing (-> enforce_limits), issued by ONE library call (icar_hip_substep) with the streaming kernels that do
    not depend on the two heavy ones beside them (icar_amd/csrc/timestep.hip has the table).  Same launches, same operands,
    same results as the plain sequence.  forced = [(member, force_boundaries), ...] with uploaded dqdt (domain.set_dqdt)."""
    domain.configure(options, forced=forced or (), diagnostics=diagnostics, prefetch_dt=prefetch_dt)
    check(lib().icar_hip_substep(domain.ctx, float(dt), int(bool(enforce))), "icar_hip_substep")


def step(domain, end_time, options, forced=None, diagnostics=True):
    """time_step.f90:440-551 for configurations 1-4 (rad/lsm/pbl/cu are no-ops there): the operator-split loop
         update_dt -> diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve
         -> advect -> apply_forcing -> enforce_limits (last two sub-steps)
    until the model clock reaches end_time, in one library call (icar_hip_step).  Returns the number of sub-steps taken."""
    domain.configure(options, forced=forced or (), diagnostics=diagnostics, prefetch_dt=True)
    n = ctypes.c_int()
    check(lib().icar_hip_step(domain.ctx, float(end_time), ctypes.byref(n)), "icar_hip_step")
    return n.value


def step_n(domain, nsteps, options, forced=None, diagnostics=True):
    """nsteps passes of update_dt -> substep -> clock += dt in one library call (icar_hip_step_n); returns the last dt."""
    domain.configure(options, forced=forced or (), diagnostics=diagnostics, prefetch_dt=True)
    dt = ctypes.c_double()
    check(lib().icar_hip_step_n(domain.ctx, int(nsteps), ctypes.byref(dt)), "icar_hip_step_n")
    return dt.value
