"""the NetCDF-classic output writer mirrors src/io/output_obj.f90 (names, dimension order, attributes,
time axis).  The reference's writer cannot run here (no NetCDF-Fortran): PARITY UNPINNED; checked is the format contract
read off the source -- CDF-1 magic, (time, level, lat_y, lon_x) order = reshape(order=[1,3,2]) of data_3d(i,k,j),
staggered dimensions, REAL(8) accumulators, appending records."""
import datetime
import numpy as np
from icar_amd.output import output_t, read_file, output_filename


class FakeDomain:
    def __init__(self, nx, ny, nz, seed=0):
        r = np.random.default_rng(seed)
        self.f = {"water_vapor": r.random((ny, nz, nx), np.float32), "potential_temperature": 300 + r.random((ny, nz, nx), np.float32),
                  "u": r.random((ny, nz, nx + 1), np.float32), "v": r.random((ny + 1, nz, nx), np.float32),
                  "accumulated_precipitation": r.random((ny, nx)), "z": r.random((ny, nz, nx), np.float32)}

    def get(self, name):
        return self.f[name]


def test_output_file_layout_and_append(tmp_path):
    d = FakeDomain(7, 5, 3)
    o = output_t(image=3, version="test")
    o.add_attribute("dx", 2000.0)
    o.add_variables(["water_vapor", "potential_temperature", "u", "v", "precipitation", "z"])
    when = datetime.datetime(2000, 1, 2, 3, 4, 5)
    fn = str(tmp_path / output_filename("icar_out_", 3, when))
    assert fn.endswith("icar_out_000003_2000-01-02_03-04-05.nc")
    o.save_file(d, fn, 1, 51545.5)
    assert open(fn, "rb").read(4) == b"CDF\x01"                      # classic format, like nf90_create(NF90_CLOBBER)
    first_qv = d.f["water_vapor"].copy()
    d.f["water_vapor"] = d.f["water_vapor"] + 1
    o.save_file(d, fn, 2, 51545.75)                                   # second record appended to the same file
    r = read_file(fn)
    assert r["_dimensions"] == {"level": 3, "lat_y": 5, "lon_x": 7, "time": None, "lon_u": 8, "lat_v": 6}
    assert r["_dims_qv"] == ("time", "level", "lat_y", "lon_x") and r["_dims_u"] == ("time", "level", "lat_y", "lon_u")
    assert r["_dims_v"] == ("time", "level", "lat_v", "lon_x") and r["_dims_precipitation"] == ("time", "lat_y", "lon_x")
    assert r["_dims_z"] == ("level", "lat_y", "lon_x")
    assert r["qv"].shape == (2, 3, 5, 7) and r["precipitation"].dtype == np.float64
    # (j,k,i) on the device == data_3d(i,k,j); the file holds (level, lat, lon)
    assert np.array_equal(r["qv"][0], first_qv.transpose(1, 0, 2)) and np.array_equal(r["qv"][1], first_qv.transpose(1, 0, 2) + 1)
    assert np.array_equal(r["time"], [51545.5, 51545.75])
    assert r["_attrs_time"]["units"] == b"days since 1858-11-17 00:00:00"
    assert r["_attrs_qv"]["standard_name"] == b"mass_fraction_of_water_vapor_in_air" and r["_attrs_qv"]["units"] == b"kg kg-1"
    assert r["_attributes"]["Conventions"] == b"CF-1.6" and int(r["_attributes"]["image"]) == 3 and r["_attributes"]["dx"] == b"2000.0"
