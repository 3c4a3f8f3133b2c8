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


def test_restart_reads_back_what_output_wrote(tmp_path):
    """restart.f90:22-81 + :83-100: every dataset variable of record `restart_step_in_file` goes back into the domain in
    data_3d(i,k,j) order; a file from another decomposition is refused; the image filename has its hour field zeroed."""
    import pytest
    from icar_amd.restart import read_restart_data, restart_model, get_image_filename
    from icar_amd.options import options_t

    class Dom(FakeDomain):
        def shape(self, fid): return self.f[fid].shape
        @staticmethod
        def fid(name): return name
        def set(self, name, a):
            assert a.dtype == self.f[name].dtype or name != "accumulated_precipitation"
            self.f[name] = np.array(a)

    d = Dom(9, 6, 4, seed=1)
    o = output_t(image=1)
    o.add_variables(["water_vapor", "potential_temperature", "u", "v", "precipitation", "z"])
    fn = str(tmp_path / "r.nc")
    rec1 = {k: v.copy() for k, v in d.f.items()}
    o.save_file(d, fn, 1, 51545.0)
    for k in d.f: d.f[k] = d.f[k] * 2
    rec2 = {k: v.copy() for k, v in d.f.items()}
    o.save_file(d, fn, 2, 51545.25)
    fresh = Dom(9, 6, 4, seed=7)
    read_restart_data(fresh, o, fn, 1)
    for k in rec1:
        want = rec1["z"] if k == "z" else rec1[k]
        assert np.array_equal(fresh.f[k], want), k
    opt = options_t(); opt.parameters.restart_file = fn; opt.parameters.restart_step_in_file = 2
    restart_model(fresh, o, opt)
    for k in rec2:
        want = rec1["z"] if k == "z" else rec2[k]            # z has no time dimension: written once, at creation
        assert np.array_equal(fresh.f[k], want), k
    with pytest.raises(Exception, match="does not match"):
        read_restart_data(Dom(8, 6, 4), o, fn, 1)
    when = datetime.datetime(2010, 10, 3, 14, 30, 0)
    assert get_image_filename(12, "restart/icar_rst_", when) == "restart/icar_rst_000012_2010-10-03_00-30-00.nc"
