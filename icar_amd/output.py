"""output_t mirror (src/io/output_h.f90, output_obj.f90; metadata src/io/default_output_metadata.f90): NetCDF classic
(CDF-1, what nf90_create(NF90_CLOBBER) produces) output files with the reference's variable names, dimension names and
order, attributes and time axis, so that tools written for ICAR output read them unchanged.  SURVEY.md 8(f) row 2.

The reference stores data_3d(i,k,j) as reshape(order=[1,3,2]) = (lon_x, lat_y, level[, time]) in Fortran order
(output_obj.f90:423), i.e. (time, level, lat_y, lon_x) in the file's C order.  Written with scipy.io.netcdf_file
(pure Python, NetCDF-3): this is host-side I/O at the output boundary, not part of the device path."""
import datetime
import numpy as np

# kVARS name -> (file variable name, dimension set, attributes)      default_output_metadata.f90
_T3 = ("time", "level", "lat_y", "lon_x"); _T3U = ("time", "level", "lat_y", "lon_u"); _T3V = ("time", "level", "lat_v", "lon_x")
_S3 = ("level", "lat_y", "lon_x"); _T2 = ("time", "lat_y", "lon_x"); _S2 = ("lat_y", "lon_x")
_LL = ("coordinates", "lat lon")
METADATA = {
    "u": ("u", _T3U, [("standard_name", "grid_eastward_wind"), ("long_name", "Grid relative eastward wind"), ("units", "m s-1"), ("coordinates", "u_lat u_lon")]),
    "v": ("v", _T3V, [("standard_name", "grid_northward_wind"), ("long_name", "Grid relative northward wind"), ("units", "m s-1"), ("coordinates", "v_lat v_lon")]),
    "w": ("w_grid", _T3, [("non_standard_name", "grid_upward_air_velocity"), ("long_name", "Vertical wind"), ("description", "Vertical wind relative to the grid"), ("units", "m s-1"), _LL]),
    "w_real": ("w", _T3, [("standard_name", "upward_air_velocity"), ("long_name", "Vertical wind"), ("description", "Vertical wind including u/v"), ("units", "m s-1"), _LL]),
    "pressure": ("pressure", _T3, [("standard_name", "air_pressure"), ("long_name", "Pressure"), ("units", "Pa"), _LL]),
    "potential_temperature": ("potential_temperature", _T3, [("standard_name", "air_potential_temperature"), ("long_name", "Potential Temperature"), ("units", "K"), _LL]),
    "temperature": ("temperature", _T3, [("standard_name", "air_temperature"), ("long_name", "Temperature"), ("units", "K"), _LL]),
    "water_vapor": ("qv", _T3, [("standard_name", "mass_fraction_of_water_vapor_in_air"), ("long_name", "Water Vapor Mixing Ratio"), ("units", "kg kg-1"), _LL]),
    "cloud_water": ("qc", _T3, [("standard_name", "cloud_liquid_water_mixing_ratio"), ("units", "kg kg-1"), _LL]),
    "cloud_ice": ("qi", _T3, [("standard_name", "cloud_ice_mixing_ratio"), ("units", "kg kg-1"), _LL]),
    "rain_in_air": ("qr", _T3, [("standard_name", "mass_fraction_of_rain_in_air"), ("units", "kg kg-1"), _LL]),
    "snow_in_air": ("qs", _T3, [("standard_name", "mass_fraction_of_snow_in_air"), ("units", "kg kg-1"), _LL]),
    "graupel_in_air": ("qg", _T3, [("standard_name", "mass_fraction_of_graupel_in_air"), ("units", "kg kg-1"), _LL]),
    "ice_number_concentration": ("ni", _T3, [("non_standard_name", "number_concentration_of_ice_crystals_in_air"), ("units", "cm-3"), _LL]),
    "rain_number_concentration": ("nr", _T3, [("non_standard_name", "number_concentration_of_rain_particles_in_air"), ("units", "cm-3"), _LL]),
    "exner": ("exner", _T3, [("non_standard_name", "exner_function_result"), ("units", "K K-1"), _LL]),
    "density": ("density", _T3, [("standard_name", "air_density"), ("units", "kg m-3"), _LL]),
    "nsquared": ("nsquared", _T3, [("standard_name", "square_of_brunt_vaisala_frequency_in_air"), ("long_name", "Burnt Vaisala frequency squared"), ("units", "s-2"), _LL]),
    "z": ("z", _S3, [("standard_name", "height_above_reference_ellipsoid"), ("units", "m"), _LL]),
    "dz": ("dz", _S3, [("non_standard_name", "layer_thickness"), ("units", "m"), _LL]),
    "precipitation": ("precipitation", _T2, [("standard_name", "precipitation_amount"), ("units", "kg m-2"), _LL]),
    "snowfall": ("snowfall", _T2, [("standard_name", "snowfall_amount"), ("units", "kg m-2"), _LL]),
    "graupel": ("graupel", _T2, [("standard_name", "graupel_amount"), ("units", "kg m-2"), _LL]),
    "surface_pressure": ("psfc", _T2, [("standard_name", "surface_air_pressure"), ("long_name", "Surface Pressure"), ("units", "Pa"), _LL]),
}
# kVARS name -> domain_t member (icar_amd.domain / _fields.NAMES)
MEMBER = {"u": "u", "v": "v", "w": "w", "w_real": "w_real", "pressure": "pressure", "potential_temperature": "potential_temperature",
          "temperature": "temperature", "water_vapor": "water_vapor", "cloud_water": "cloud_water_mass", "cloud_ice": "cloud_ice_mass",
          "rain_in_air": "rain_mass", "snow_in_air": "snow_mass", "graupel_in_air": "graupel_mass",
          "ice_number_concentration": "cloud_ice_number", "rain_number_concentration": "rain_number", "exner": "exner",
          "density": "density", "nsquared": "nsquared", "z": "z", "dz": "dz_mass", "precipitation": "accumulated_precipitation",
          "snowfall": "accumulated_snowfall", "graupel": "graupel", "surface_pressure": "surface_pressure"}
MJD_UNITS = "days since 1858-11-17 00:00:00"


def output_filename(prefix, image, when):
    """driver.f90:94-97: <output_file><image, 6 digits>_<YYYY-MM-DD_hh-mm-ss>.nc"""
    return f"{prefix}{image:06d}_{when.strftime('%Y-%m-%d_%H-%M-%S')}.nc"


class output_t:
    def __init__(self, image=1, version="icar_amd"):
        self.variables = []                       # kVARS names, in the order added (output_obj.f90:22-36)
        self.attributes = []
        self.image = image
        self.version = version

    def add_attribute(self, name, value):
        self.attributes.append((name, str(value)))

    def add_variables(self, var_list, domain=None):
        """output_obj.f90:80-: every requested kVARS entry that has output metadata."""
        for n in var_list:
            if n not in METADATA:
                raise KeyError(f"no output metadata for {n}")
            if n not in self.variables:
                self.variables.append(n)

    def save_file(self, domain, filename, current_step, time_mjd, calendar="gregorian"):
        """save_file (output_obj.f90:41-78): create the file (or reopen it) and store the variables at record
        `current_step` (1-based) with the time in modified Julian days (save_data :380-460)."""
        import os
        from scipy.io import netcdf_file
        creating = not os.path.exists(filename)
        f = netcdf_file(filename, "w" if creating else "a", version=1)
        try:
            arrays = {}
            if creating:
                f.createDimension("time", None)              # the record dimension has to come first in a classic file
            for n in self.variables:
                name, dims, attrs = METADATA[n]
                a = np.asarray(domain.get(MEMBER[n]))
                a = a.transpose(1, 0, 2) if a.ndim == 3 else a            # (j,k,i) -> (level, lat, lon)
                arrays[n] = a
                if creating:
                    space = dims[1:] if dims[0] == "time" else dims
                    for dname, dlen in zip(space, a.shape):
                        if dname not in f.dimensions:
                            f.createDimension(dname, int(dlen))
                    v = f.createVariable(name, "d" if a.dtype == np.float64 else "f", dims)
                    for k, val in attrs:
                        setattr(v, k, val)
            if creating:
                t = f.createVariable("time", "d", ("time",))
                t.standard_name = "time"; t.calendar = calendar; t.units = MJD_UNITS; t.UTCoffset = "0"
                f.Conventions = "CF-1.6"
                f.title = "Intermediate Complexity Atmospheric Research (ICAR) model output"
                f.institution = "National Center for Atmospheric Research"
                f.references = ("Gutmann et al. 2016: The Intermediate Complexity Atmospheric Model (ICAR). "
                                "J.Hydrometeor. doi:10.1175/JHM-D-15-0155.1, 2016.")
                f.contact = "Ethan Gutmann : gutmann@ucar.edu"
                f.git = self.version
                for k, val in self.attributes:
                    setattr(f, k, val)
                from ._netcdf import FORMAT_NOTE
                f.format_note = FORMAT_NOTE
                f.history = "Created:" + datetime.datetime.now().strftime("%Y/%m/%d %H:%M:%S")
                f.image = np.int32(self.image)
            rec = current_step - 1
            for n in self.variables:
                name, dims, _ = METADATA[n]
                v = f.variables[name]
                if dims[0] == "time":
                    v[rec] = arrays[n]
                elif creating:
                    v[:] = arrays[n]
            f.variables["time"][rec] = float(time_mjd)
        finally:
            f.close()


def read_file(filename):
    """{file variable name: array} (+ "_dimensions", "_attributes") -- restart-style reader of the files above."""
    from ._netcdf import open_classic
    out = {}
    with open_classic(filename) as f:
        out["_dimensions"] = {k: (None if v is None else int(v)) for k, v in f.dimensions.items()}
        out["_attributes"] = {k: getattr(f, k) for k in f._attributes}
        for k, v in f.variables.items():
            a = np.array(v[:])
            out[k] = a.astype(a.dtype.newbyteorder("="))            # classic files are big-endian
            out["_dims_" + k] = tuple(v.dimensions)
            out["_attrs_" + k] = dict(v._attributes)
    return out
