"""wind module mirror (src/physics/wind.f90): update_winds / balance_uvw for the wind types on the device path."""
import ctypes
from .capi import lib, check, IcarHipError
from .linear_winds import linear_perturb

from . import _fields as F

kWIND_LINEAR = 1                 # icar_constants.f90:368-377 (windtype)
kCONSERVE_MASS = 2
kITERATIVE_WINDS = 3
kLINEAR_ITERATIVE_WINDS = 5


def balance_uvw(domain, update=False):
    """wind.f90:81-169: w (or its dqdt_3d when update) from the horizontal divergence of u, v."""
    fn = lib().icar_hip_balance_uvw_update if update else lib().icar_hip_balance_uvw
    check(fn(domain.ctx, ctypes.c_float(domain.dx)), "balance_uvw")


def make_winds_grid_relative(domain, update=False):
    """wind.f90:236-287 on domain%u / domain%v (their dqdt_3d when update).  domain%sintheta / costheta are what init_winds
    (wind.f90:512-590) derives from the lat / lon grid; a domain that has none is an unrotated grid (sin 0, cos 1)."""
    if not getattr(domain, "_has_theta", False):
        import numpy as np
        domain.set("sintheta", np.zeros((domain.ny, domain.nx), np.float64))
        domain.set("costheta", np.ones((domain.ny, domain.nx), np.float64))
    check(lib().icar_hip_make_winds_grid_relative(domain.ctx, int(update)), "make_winds_grid_relative")


def exchange_uv(domain, update=False):
    """domain%u%exchange_u(); domain%v%exchange_v() (on the dqdt_3d arrays when iterative_winds swapped them in)."""
    if getattr(domain, "comm", None) is not None:
        domain.comm.exchange_uv(domain, F.U, F.V, which=1 if update else 0)


def iterative_winds(domain, options, update=False):
    """wind.f90:371-498: Jacobi-like removal of the 3-D divergence from u, v with w pinned to zero at the model top.
    Same control flow as the reference; on one image the loop is a single device call."""
    n = int(options.parameters.wind_iterations) + 1          # do it = 0, wind_iterations
    dx = ctypes.c_float(domain.dx)
    exchange_uv(domain, update)
    balance_uvw(domain, update)
    check(lib().icar_hip_iterative_winds_correct_w(domain.ctx, int(update)), "iterative_winds")
    if getattr(domain, "comm", None) is None or not domain.comm.peers:
        check(lib().icar_hip_iterative_winds_sweep(domain.ctx, dx, n, int(update)), "iterative_winds")
        return
    for _ in range(n):
        check(lib().icar_hip_iterative_winds_sweep(domain.ctx, dx, 1, int(update)), "iterative_winds")
        exchange_uv(domain, update)


def update_winds(domain, options):
    """wind.f90:289-360 for every windtype: 0, kWIND_LINEAR, kCONSERVE_MASS, kITERATIVE_WINDS, kLINEAR_ITERATIVE_WINDS.  First call: linear_perturb on u, v then balance_uvw on the
    winds; every later call (a new forcing step has put the next winds into dqdt_3d) the same on the tendencies.  Both
    branches start with make_winds_grid_relative (:300, :338).
    setup_linwinds(domain, options, global_terrain) must have been called when windtype == kWIND_LINEAR."""
    wt = options.physics.windtype
    if wt not in (0, kWIND_LINEAR, kCONSERVE_MASS, kITERATIVE_WINDS, kLINEAR_ITERATIVE_WINDS):
        raise IcarHipError(f"update_winds: unknown windtype {wt}")
    first = not getattr(domain, "_winds_initialised", False)
    make_winds_grid_relative(domain, update=not first)                                   # wind.f90:300 / :338
    if wt in (kWIND_LINEAR, kLINEAR_ITERATIVE_WINDS):
        linear_perturb(domain, options, options.lt_options.vert_smooth, False, options.parameters.advect_density, update=not first)
    if wt == kCONSERVE_MASS:
        # wind.f90:301-306 / :333-338: the host has uploaded zr_u / zr_v (zfr_* with use_terrain_difference) as "zr_u" / "zr_v"
        check(lib().icar_hip_mass_conservative_acceleration(domain.ctx, int(not first)), "mass_conservative_acceleration")
    if wt in (kITERATIVE_WINDS, kLINEAR_ITERATIVE_WINDS):
        iterative_winds(domain, options, update=not first)
    balance_uvw(domain, update=not first)
    domain._winds_initialised = True
