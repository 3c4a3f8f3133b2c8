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
    """domain%u%exchange_u(); domain%v%exchange_v() (on the dqdt_3d arrays when iterative_winds swapped them in): one call
    into the library, which moves one message per neighbour over the context's communicator (comm.hip)."""
    check(lib().icar_hip_exchange_uv(domain.ctx, int(getattr(domain.comm, "halo", 1) if getattr(domain, "comm", None) is not None else 1),
                                     int(update)), "exchange_uv")


def update_winds(domain, options):
    """wind.f90:289-369 for every windtype (0, kWIND_LINEAR, kCONSERVE_MASS, kITERATIVE_WINDS, kLINEAR_ITERATIVE_WINDS) as ONE
    library call (icar_hip_update_winds: make_winds_grid_relative -> linear_perturb -> mass_conservative_acceleration ->
    iterative_winds with its exchange_u / exchange_v per sweep -> balance_uvw).  First call: on u, v, w; every later call (a
    new forcing step has put the next winds into dqdt_3d): on the tendencies.
    setup_linwinds(domain, options, global_terrain) must have been called for the linear wind types; domain%sintheta /
    costheta are what init_winds (wind.f90:512-590) derives from the lat / lon grid -- a domain that has none is an unrotated grid."""
    wt = options.physics.windtype
    if wt not in (0, kWIND_LINEAR, kCONSERVE_MASS, kITERATIVE_WINDS, kLINEAR_ITERATIVE_WINDS):
        raise IcarHipError(f"update_winds: unknown windtype {wt}")
    if wt in (kWIND_LINEAR, kLINEAR_ITERATIVE_WINDS) and not getattr(domain, "_linwinds_ready", False):
        raise IcarHipError("linear_perturb: call setup_linwinds(domain, options, global_terrain) first")
    first = not getattr(domain, "_winds_initialised", False)
    if not getattr(domain, "_has_theta", False):
        import numpy as np
        domain.set("sintheta", np.zeros((domain.ny, domain.nx), np.float64))
        domain.set("costheta", np.ones((domain.ny, domain.nx), np.float64))
    halo = int(getattr(domain.comm, "halo", 1)) if getattr(domain, "comm", None) is not None else 1
    check(lib().icar_hip_update_winds(domain.ctx, int(wt), int(options.parameters.wind_iterations), ctypes.c_float(domain.dx), halo,
                                      0 if first else 1), "update_winds")
    domain._winds_initialised = True


def iterative_winds(domain, options, update=False):
    """wind.f90:371-498 alone (update_winds runs it inside the library): the reference's control flow, call by call."""
    n = int(options.parameters.wind_iterations) + 1          # do it = 0, wind_iterations
    dx = ctypes.c_float(domain.dx)
    exchange_uv(domain, update)
    balance_uvw(domain, update)
    check(lib().icar_hip_iterative_winds_correct_w(domain.ctx, int(update)), "iterative_winds")
    for _ in range(n):
        check(lib().icar_hip_iterative_winds_sweep(domain.ctx, dx, 1, int(update)), "iterative_winds")
        exchange_uv(domain, update)
