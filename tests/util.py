"""Shared helpers for the parity tests."""
import numpy as np
from icar_amd import ideal
from icar_amd.grid import grid_t
from icar_amd.options import options_t

SCALARS = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel",
           "ice_number", "rain_number"]
MEMBER = {"water_vapor": "water_vapor", "cloud_water": "cloud_water_mass", "rain": "rain_mass", "snow": "snow_mass",
          "potential_temperature": "potential_temperature", "cloud_ice": "cloud_ice_mass", "graupel": "graupel_mass",
          "ice_number": "cloud_ice_number", "rain_number": "rain_number"}
KVAR = {"water_vapor": "water_vapor", "cloud_water": "cloud_water", "rain": "rain_in_air", "snow": "snow_in_air",
        "potential_temperature": "potential_temperature", "cloud_ice": "cloud_ice", "graupel": "graupel_in_air",
        "ice_number": "ice_number_concentration", "rain_number": "rain_number_concentration"}


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int32), np.ascontiguousarray(b).view(np.int32))


def nbitdiff(a, b):
    return int((np.ascontiguousarray(a).view(np.int32) != np.ascontiguousarray(b).view(np.int32)).sum())


def single_image_domain(case, device=0):
    from icar_amd.domain import domain_t
    g = grid_t().set_grid_dimensions(case["nx"], case["ny"], case["nz"], 1, 1)
    d = domain_t(g, device=device, dx=float(case["dx"]))
    d.load_case(case)
    return d


def adv_args(c):
    return (c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
            c["advection_dz"], c["dz_levels"], float(c["dx"]))
