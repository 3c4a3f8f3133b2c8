"""wind module mirror (src/physics/wind.f90): update_winds / balance_uvw for the wind types on the device path."""
import ctypes
from .capi import lib, check, IcarHipError
from .linear_winds import linear_perturb

kWIND_LINEAR = 1                 # icar_constants.f90:368 (windtype)


def balance_uvw(domain, update=False):
    """wind.f90:81-169: w (or its dqdt_3d when update) from the horizontal divergence of u, v."""
    fn = lib().icar_hip_balance_uvw_update if update else lib().icar_hip_balance_uvw
    check(fn(domain.ctx, ctypes.c_float(domain.dx)), "balance_uvw")


def update_winds(domain, options):
    """wind.f90:289-360 for windtype 0 and kWIND_LINEAR.  First call: linear_perturb on u, v then balance_uvw on the
    winds; every later call (a new forcing step has put the next winds into dqdt_3d) the same on the tendencies.
    make_winds_grid_relative (rotation by sintheta / costheta) belongs to the forcing reader and is not on this path;
    setup_linwinds(domain, options, global_terrain) must have been called when windtype == kWIND_LINEAR."""
    wt = options.physics.windtype
    if wt not in (0, kWIND_LINEAR):
        raise IcarHipError("update_winds: only windtype 0 and kWIND_LINEAR are on the device path")
    first = not getattr(domain, "_winds_initialised", False)
    if wt == kWIND_LINEAR:
        linear_perturb(domain, options, options.lt_options.vert_smooth, False, options.parameters.advect_density, update=not first)
    balance_uvw(domain, update=not first)
    domain._winds_initialised = True
