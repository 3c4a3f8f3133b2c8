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
        domain.mp_state = dict(last_model_time=-999.0)       # the module SAVE variables live on the domain object


def mp_tiles(its, ite, jts, jte, halo=0, subset=0):
    """process_halo / subset tile bookkeeping (mp_driver.f90:609-658, :728-737), via the C ABI."""
    t = ((ctypes.c_int * 4) * 4)()
    n = lib().icar_hip_mp_tiles(its, ite, jts, jte, int(halo), int(subset), t)
    return [tuple(t[i]) for i in range(n)]


def _process_subdomain(domain, options, dt, its, ite, jts, jte, kts, kte):
    g = domain.grid
    mp_ = options.physics.microphysics
    if ite < its or jte < jts:
        return
    if mp_ == kMP_SB04:
        check(lib().icar_hip_mp_simple(domain.ctx, ctypes.c_float(dt), its, ite, jts, jte, kts, kte, None),
              "icar_hip_mp_simple")
    elif mp_ == kMP_WSM6:
        check(lib().icar_hip_wsm6(domain.ctx, ctypes.c_float(dt), its, ite, jts, jte, kts, kte), "icar_hip_wsm6")
    elif mp_ == kMP_WSM3:
        check(lib().icar_hip_wsm3(domain.ctx, ctypes.c_float(dt), its, ite, jts, jte, kts, kte), "icar_hip_wsm3")
    elif mp_ == kMP_THOMPSON:
        check(lib().icar_hip_thompson(domain.ctx, ctypes.c_float(dt), its, ite, jts, jte, kts, kte,
                                      g.ids, g.ide, g.jds, g.jde, g.kds, g.kde), "icar_hip_thompson")


def mp(domain, options, dt_in, halo=None, subset=None):
    """mp_driver.f90:673-772 including the update_interval gating (:698-713)."""
    if options.physics.microphysics == 0:
        return
    st = domain.mp_state
    upd = float(options.mp_options.update_interval)
    now = domain.model_time_seconds
    if st["last_model_time"] == -999.0:
        st["last_model_time"] = now - max(upd, float(dt_in))
    if ((now + dt_in) - st["last_model_time"]) >= upd:
        mp_dt = now - st["last_model_time"]
        if halo is None:
            st["last_model_time"] = now
        g = domain.grid
        kte = g.kte
        if options.mp_options.top_mp_level > 0:
            kte = min(kte, options.mp_options.top_mp_level)
        if subset is not None:
            for (a, b, c, d) in mp_tiles(g.its, g.ite, g.jts, g.jte, subset=subset):
                _process_subdomain(domain, options, mp_dt, a, b, c, d, g.kts, kte)
        if halo is not None:
            tiles = mp_tiles(g.its, g.ite, g.jts, g.jte, halo=halo)
            h = int(halo)
            # a tile narrower than 2*halo makes the west/east (or south/north) strips overlap: the reference then runs
            # those columns once per strip, one strip after the other -- a single batched launch would race on them
            overlapping = (g.ite - g.its + 1 < 2 * h) or (g.jte - g.jts + 1 < 2 * h)
            if options.physics.microphysics in (kMP_THOMPSON, kMP_SB04, kMP_WSM6) and not overlapping:
                # process_halo's four strips in one launch (icar_hip_thompson_tiles / icar_hip_mp_simple_tiles / icar_hip_wsm6_tiles)
                tiles = [t for t in tiles if t[1] >= t[0] and t[3] >= t[2]]
                arr = ((ctypes.c_int * 4) * len(tiles))(*[(ctypes.c_int * 4)(*t) for t in tiles])
                if tiles and options.physics.microphysics == kMP_THOMPSON:
                    check(lib().icar_hip_thompson_tiles(domain.ctx, ctypes.c_float(mp_dt), len(tiles), arr, g.kts, kte,
                                                        g.ids, g.ide, g.jds, g.jde, g.kds, g.kde), "icar_hip_thompson_tiles")
                elif tiles and options.physics.microphysics == kMP_WSM6:
                    check(lib().icar_hip_wsm6_tiles(domain.ctx, ctypes.c_float(mp_dt), len(tiles), arr, g.kts, kte), "icar_hip_wsm6_tiles")
                elif tiles:
                    check(lib().icar_hip_mp_simple_tiles(domain.ctx, ctypes.c_float(mp_dt), len(tiles), arr, g.kts, kte, None),
                          "icar_hip_mp_simple_tiles")
            else:
                for (a, b, c, d) in tiles:
                    _process_subdomain(domain, options, mp_dt, a, b, c, d, g.kts, kte)
        if halo is None and subset is None:
            _process_subdomain(domain, options, mp_dt, g.its, g.ite, g.jts, g.jte, g.kts, kte)


def mp_finish(options, domain=None):
    if domain is not None:
        domain.mp_state = dict(last_model_time=-999.0)
