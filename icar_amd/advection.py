"""advection driver mirror (src/physics/advection_driver.f90): adv_var_request / adv_init / advect."""
import ctypes
import numpy as np
from .capi import lib, check
from .constants import ADVECTION_ORDER, KVARS, kADV_UPWIND, kADV_MPDATA


def adv_var_request(options):
    """advection_driver.f90:39-49 -> upwind_var_request / mpdata request: u, v, w, dz."""
    if options.physics.advection in (kADV_UPWIND, kADV_MPDATA):
        options.alloc_vars(["u", "v", "w", "dz_interface"])
        options.restart_vars(["u", "v", "w", "dz_interface"])


def adv_init(domain, options):
    """advection_driver.f90:22-37: module state is the device context; nothing else to set up."""
    if options.physics.advection not in (0, kADV_UPWIND, kADV_MPDATA):
        raise ValueError("unknown advection option")


def advected_field_ids(options):
    """Field ids in the reference's fixed dispatch order (adv_mpdata.f90:512-522)."""
    return [KVARS[n][0] for n in ADVECTION_ORDER if options.vars_to_advect.get(n, 0) > 0]


def setup_winds(domain, options, dt):
    """setup_module_winds (advect.f90:306-351 / adv_mpdata.f90:496-506): U_m, V_m, W_m (, W_m/dz) of this step's dt.  The library
    remembers what they were set up for; advect() redoes the setup only when that no longer matches (or the winds moved)."""
    scheme = options.physics.advection
    dens = int(bool(options.parameters.advect_density))
    check(lib().icar_hip_setup_winds(domain.ctx, scheme, ctypes.c_float(dt), ctypes.c_float(domain.dx), dens),
          "icar_hip_setup_winds")


def advect(domain, options, dt):
    """advection_driver.f90:51-77: advect every scalar with vars_to_advect>0 over one step dt (icar_hip_advect_step)."""
    if options.physics.advection not in (kADV_UPWIND, kADV_MPDATA):
        return
    domain.configure(options)
    check(lib().icar_hip_advect_step(domain.ctx, float(dt)), "icar_hip_advect_step")
