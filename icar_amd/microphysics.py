"""microphysics driver mirror (src/physics/mp_driver.f90): mp_var_request / mp_init / mp / mp_finish."""
import ctypes
from .capi import lib, check
from .constants import kMP_THOMPSON, kMP_SB04, kMP_WSM3, kMP_WSM6


def mp_var_request(options):
    """mp_driver.f90:200-229 (mp_simple.f90:104-126, mp_driver.f90:115-140 for Thompson)."""
    mp = options.physics.microphysics
    if mp == kMP_SB04:
        options.alloc_vars(["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water",
                            "rain_in_air", "snow_in_air", "precipitation", "snowfall", "dz"])
        options.advect_vars(["potential_temperature", "water_vapor", "cloud_water", "rain_in_air", "snow_in_air"])
        options.restart_vars(["pressure", "potential_temperature", "water_vapor", "cloud_water", "rain_in_air",
                              "snow_in_air", "precipitation", "snowfall", "dz"])
    elif mp == kMP_THOMPSON:
        options.alloc_vars(["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water",
                            "rain_in_air", "rain_number_concentration", "snow_in_air", "cloud_ice",
                            "ice_number_concentration", "graupel_in_air", "precipitation", "snowfall", "graupel", "dz"])
        options.advect_vars(["potential_temperature", "water_vapor", "cloud_water", "rain_number_concentration",
                             "snow_in_air", "cloud_ice", "rain_in_air", "ice_number_concentration", "graupel_in_air"])
        options.restart_vars(["pressure", "potential_temperature", "water_vapor", "cloud_water", "rain_in_air",
                              "rain_number_concentration", "snow_in_air", "cloud_ice", "ice_number_concentration",
                              "graupel_in_air", "precipitation", "snowfall", "graupel", "dz"])
    elif mp == kMP_WSM6:                                # mp_driver.f90:146-172
        options.alloc_vars(["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain_in_air",
                            "snow_in_air", "cloud_ice", "dz", "snowfall", "precipitation", "graupel", "graupel_in_air"])
        options.advect_vars(["potential_temperature", "water_vapor", "cloud_water", "snow_in_air", "cloud_ice", "rain_in_air",
                             "graupel_in_air"])
        options.restart_vars(["pressure", "potential_temperature", "water_vapor", "cloud_water", "rain_in_air", "snow_in_air",
                              "precipitation", "snowfall", "graupel", "dz", "snow_in_air", "cloud_ice", "rain_in_air", "graupel_in_air"])
    elif mp == kMP_WSM3:                                # mp_driver.f90:174-198
        options.alloc_vars(["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain_in_air",
                            "dz", "snowfall", "precipitation"])
        options.advect_vars(["potential_temperature", "water_vapor", "cloud_water", "rain_in_air"])
        options.restart_vars(["pressure", "potential_temperature", "water_vapor", "cloud_water", "rain_in_air", "snow_in_air",
                              "precipitation", "snowfall", "graupel", "dz", "rain_in_air"])
    elif mp == 0 and options.parameters.ideal:
        options.advect_vars(["water_vapor"])            # mp_driver.f90:223-226


def mp_init(options, domain=None):
    """mp_driver.f90:74-113: build the scheme's tables (Thompson) and reset the update counter."""
    if options.physics.microphysics == kMP_THOMPSON:
        if domain is None:
            raise ValueError("mp_init(options, domain): the Thompson tables live in the domain's device context")
        p, f = options.mp_options.as_arrays()
        check(lib().icar_hip_thompson_init(domain.ctx, p.ctypes.data_as(ctypes.c_void_p),
                                           f.ctypes.data_as(ctypes.c_void_p)), "icar_hip_thompson_init")
    if options.physics.microphysics == kMP_WSM6:
        if domain is None:
            raise ValueError("mp_init(options, domain): the WSM6 constants live in the domain's device context")
        check(lib().icar_hip_wsm6_init(domain.ctx), "icar_hip_wsm6_init")
    if options.physics.microphysics == kMP_WSM3:
        if domain is None:
            raise ValueError("mp_init(options, domain): the WSM3 constants live in the domain's device context")
        check(lib().icar_hip_wsm3_init(domain.ctx), "icar_hip_wsm3_init")
    if domain is not None:
        check(lib().icar_hip_mp_reset(domain.ctx), "icar_hip_mp_reset")      # the module SAVE variable last_model_time lives in the context


def mp_tiles(its, ite, jts, jte, halo=0, subset=0):
    """process_halo / subset tile bookkeeping (mp_driver.f90:609-658, :728-737), via the C ABI."""
    t = ((ctypes.c_int * 4) * 4)()
    n = lib().icar_hip_mp_tiles(its, ite, jts, jte, int(halo), int(subset), t)
    return [tuple(t[i]) for i in range(n)]


def mp(domain, options, dt_in, halo=None, subset=None):
    """mp(domain, options, dt_in, halo, subset) (mp_driver.f90:673-772) including the update_interval gating (:698-713) and
    process_halo's strips (:609-658): icar_hip_mp, the same entry point a Fortran host calls."""
    if options.physics.microphysics == 0:
        return
    domain.configure(options)
    check(lib().icar_hip_mp(domain.ctx, float(dt_in), -1 if halo is None else int(halo), -1 if subset is None else int(subset)),
          "icar_hip_mp")


def mp_finish(options, domain=None):
    if domain is not None:
        check(lib().icar_hip_mp_reset(domain.ctx), "icar_hip_mp_reset")
