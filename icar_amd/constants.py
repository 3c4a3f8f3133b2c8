"""Constants mirrored from the reference (src/constants/icar_constants.f90)."""
from . import _fields as F

# physics selectors, icar_constants.f90:341-374
kADV_UPWIND = 1
kADV_MPDATA = 2
kMP_THOMPSON = 1
kMP_SB04 = 2
kMP_WSM6 = 4
kMP_WSM3 = 6
kDEFAULT_HALO_SIZE = 1          # icar_constants.f90:320

# advection dispatch order of mpdata()/upwind() (src/physics/adv_mpdata.f90:512-522);
# the ids double as the C-ABI field ids (include/icar_hip.h).
ADVECTION_ORDER = [
    "water_vapor", "cloud_water", "rain_in_air", "snow_in_air", "potential_temperature",
    "cloud_ice", "graupel_in_air", "ice_number_concentration", "rain_number_concentration",
    "snow_number_concentration", "graupel_number_concentration",
]

# kVARS name -> (C-ABI field id, domain_t member name)
KVARS = {
    "water_vapor": (F.WATER_VAPOR, "water_vapor"),
    "cloud_water": (F.CLOUD_WATER, "cloud_water_mass"),
    "rain_in_air": (F.RAIN, "rain_mass"),
    "snow_in_air": (F.SNOW, "snow_mass"),
    "potential_temperature": (F.POTENTIAL_TEMPERATURE, "potential_temperature"),
    "cloud_ice": (F.CLOUD_ICE, "cloud_ice_mass"),
    "graupel_in_air": (F.GRAUPEL, "graupel_mass"),
    "ice_number_concentration": (F.ICE_NUMBER, "cloud_ice_number"),
    "rain_number_concentration": (F.RAIN_NUMBER, "rain_number"),
    "snow_number_concentration": (F.SNOW_NUMBER, "snow_number"),
    "graupel_number_concentration": (F.GRAUPEL_NUMBER, "graupel_number"),
    "u": (F.U, "u"), "v": (F.V, "v"), "w": (F.W, "w"),
    "pressure": (F.PRESSURE, "pressure"), "exner": (F.EXNER, "exner"),
    "density": (F.DENSITY, "density"), "dz": (F.DZ_MASS, "dz_mass"),
    "precipitation": (F.PRECIPITATION, "accumulated_precipitation"),
    "snowfall": (F.SNOWFALL, "accumulated_snowfall"),
    "graupel": (F.GRAUPEL_ACC, "graupel"),
}
