"""Field ids of include/icar_hip.h (enum icar_hip_field)."""
WATER_VAPOR, CLOUD_WATER, RAIN, SNOW, POTENTIAL_TEMPERATURE, CLOUD_ICE, GRAUPEL = range(7)
ICE_NUMBER, RAIN_NUMBER, SNOW_NUMBER, GRAUPEL_NUMBER = 7, 8, 9, 10
N_ADVECTABLE = 11
U, V, W, PRESSURE, EXNER, DENSITY, DZ_MASS = 11, 12, 13, 14, 15, 16, 17
JACOBIAN, JACOBIAN_U, JACOBIAN_V, JACOBIAN_W, ADVECTION_DZ = 18, 19, 20, 21, 22
PRECIPITATION, SNOWFALL, GRAUPEL_ACC = 23, 24, 25
PRESSURE_INTERFACE, TEMPERATURE, TEMPERATURE_INTERFACE, U_MASS, V_MASS, W_REAL, DZDX, DZDY, SURFACE_PRESSURE = range(26, 35)
Z, NSQUARED = 35, 36
IVT, IWV, IWL, IWI = 37, 38, 39, 40
ZR_U, ZR_V = 41, 42
SINTHETA, COSTHETA = 43, 44
N_FIELDS = 45

NAMES = {
    "water_vapor": WATER_VAPOR, "cloud_water_mass": CLOUD_WATER, "rain_mass": RAIN, "snow_mass": SNOW,
    "potential_temperature": POTENTIAL_TEMPERATURE, "cloud_ice_mass": CLOUD_ICE, "graupel_mass": GRAUPEL,
    "cloud_ice_number": ICE_NUMBER, "rain_number": RAIN_NUMBER, "snow_number": SNOW_NUMBER,
    "graupel_number": GRAUPEL_NUMBER, "u": U, "v": V, "w": W, "pressure": PRESSURE, "exner": EXNER,
    "density": DENSITY, "dz_mass": DZ_MASS, "jacobian": JACOBIAN, "jacobian_u": JACOBIAN_U,
    "jacobian_v": JACOBIAN_V, "jacobian_w": JACOBIAN_W, "advection_dz": ADVECTION_DZ,
    "pressure_interface": PRESSURE_INTERFACE, "temperature": TEMPERATURE, "temperature_interface": TEMPERATURE_INTERFACE,
    "u_mass": U_MASS, "v_mass": V_MASS, "w_real": W_REAL, "dzdx": DZDX, "dzdy": DZDY, "surface_pressure": SURFACE_PRESSURE,
    "z": Z, "nsquared": NSQUARED, "ivt": IVT, "iwv": IWV, "iwl": IWL, "iwi": IWI, "zr_u": ZR_U, "zr_v": ZR_V, "sintheta": SINTHETA, "costheta": COSTHETA,
    "accumulated_precipitation": PRECIPITATION, "accumulated_snowfall": SNOWFALL, "graupel": GRAUPEL_ACC,
}
IS_2DD = {PRECIPITATION, SNOWFALL, GRAUPEL_ACC, SINTHETA, COSTHETA}
