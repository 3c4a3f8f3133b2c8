"""SURVEY 8(f) row 2, input side: the ideal-case files (variable / dimension names of the reference's generator,
tests/gen_ideal_test.py -> helpers/genNetCDF) read into the members of domain_t.  CPU-only: format contract + the reader's
interpolation against the generating formulas.  The run from these files is tests/test_gpu_ideal_run.py."""
import os
import sys
import numpy as np
import pytest
from scipy.io import netcdf_file

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_ideal_files as M
from icar_amd import ideal, ideal_io


def test_files_carry_the_reference_names_and_shapes(tmp_path):
    init, forcing = M.write(str(tmp_path))
    with netcdf_file(init, "r", mmap=False) as f:
        assert set(f.variables) == {"lat_hi", "lon_hi", "hgt_hi"}                                      # Topography.py:62-66
        assert f.variables["hgt_hi"].dimensions == ("lat", "lon") and f.variables["hgt_hi"].shape == (M.NY, M.NX)
        assert f.variables["hgt_hi"].units == b"meters MSL" and f.GRIDTYPE == b"C"
        h = f.variables["hgt_hi"][:]
        assert abs(h.max() - M.HILL_HEIGHT) < 1e-6 * M.HILL_HEIGHT and h.min() >= 0                     # one cosine hill, centred
        assert np.unravel_index(h.argmax(), h.shape) == (M.NY // 2, M.NX // 2)
    with netcdf_file(forcing, "r", mmap=False) as f:
        assert set(f.variables) == {"u", "v", "theta", "qv", "height", "z", "pressure", "temperature", "lat_m", "lon_m", "x_m", "time"}   # Forcing.py:62-74
        for n in ("u", "v", "theta", "qv", "z", "pressure", "temperature"):
            assert f.variables[n].dimensions == ("time", "level", "lat", "lon") and f.variables[n].shape == (M.NT_LO, M.NZ_LO, M.NY + 10, M.NX + 10)
        assert f.dimensions["time"] is None                                                              # unlimited_dims='time'
        z = f.variables["z"][0, :, 0, 0]
        assert np.array_equal(z, np.arange(M.NZ_LO) * M.DZ_LO)
        th = f.variables["theta"][0, :, 3, 4]
        assert abs(th[0] - 300.0) < 1e-12 and abs(th[24] - 343.0) < 1e-9 and th[30] > th[24]            # Weisman-Klemp: 300 K at 0 m, 343 K at 12 km
        p = f.variables["pressure"][0, :, 0, 0]
        assert abs(p[0] - 1e5) < 1e-6 and np.all(np.diff(p) < 0)


def test_reader_builds_the_domain_members(tmp_path):
    init, forcing = M.write(str(tmp_path))
    dzl = ideal.dz_levels(M.NZ)
    c = ideal_io.read_ideal_case(init, forcing, dzl, M.DX)
    nx, ny, nz = M.NX, M.NY, M.NZ
    assert c["terrain"].shape == (ny, nx) and c["u"].shape == (ny, nz, nx + 1) and c["v"].shape == (ny + 1, nz, nx)
    for k in ("potential_temperature", "pressure", "exner", "density", "water_vapor", "w", "jacobian", "advection_dz", "dz_mass", "z"):
        assert c[k].shape == (ny, nz, nx) and c[k].dtype == np.float32 and np.isfinite(c[k]).all(), k
    # the forcing is horizontally uniform: every column is the generating profile at that column's mass-level heights
    z = c["z"].astype(np.float64)
    zf = np.arange(M.NZ_LO) * M.DZ_LO
    th_want = np.interp(z, zf, ideal_io.calc_wk_theta(zf))
    assert np.abs(c["potential_temperature"] - th_want).max() < 2e-4
    p_want = np.exp(np.interp(z, zf, np.log(ideal_io.calc_pressure_from_sea(1e5, zf))))
    assert np.abs(c["pressure"] / p_want - 1).max() < 1e-6
    assert np.allclose(c["water_vapor"], M.QV_VAL) and np.allclose(c["u"], M.U_VAL) and np.allclose(c["v"], M.V_VAL)
    assert np.abs(c["exner"] - (c["pressure"].astype(np.float64) / 1e5) ** (287.058 / 1012.0)).max() < 1e-6
    # terrain-following levels: higher over the hill, the column depth shrinks by the jacobian
    j, i = ny // 2, nx // 2
    assert abs(z[j, 0, i] - (M.HILL_HEIGHT + 0.5 * dzl[0] * c["jacobian"][j, 0, i])) < 1e-2
    assert np.abs(c["w"]).max() > 0.05                      # flow over the hill has vertical motion (balance_uvw)
    assert np.abs(c["w"][0]).max() < 0.1 * np.abs(c["w"]).max()      # nearly flat at the domain edge


def test_netcdf4_files_are_refused_with_advice(tmp_path):
    """The reference's generators (xarray) and its own output / LUT writers produce NetCDF-4 / HDF5; this build has no HDF5
    library.  Every reader must say so -- with the conversion that makes the file readable -- instead of failing inside scipy."""
    from icar_amd.capi import IcarHipError
    from icar_amd import ideal_io, output
    from icar_amd._netcdf import HDF5_SIGNATURE, FORMAT_NOTE
    h5 = tmp_path / "init.nc"
    h5.write_bytes(HDF5_SIGNATURE + b"\0" * 512)
    ok = tmp_path / "forcing.nc"
    ideal_io.write_forcing(str(ok), 1, 4, 6, 5)
    for call in (lambda: ideal_io.read_ideal_case(str(h5), str(ok), [200.0] * 4, 1000.0),
                 lambda: output.read_file(str(h5))):
        with pytest.raises(IcarHipError) as e:
            call()
        assert "NetCDF-4" in str(e.value) and "nccopy -k classic" in str(e.value)
    junk = tmp_path / "junk.nc"; junk.write_bytes(b"not a netcdf file")
    with pytest.raises(IcarHipError):
        output.read_file(str(junk))
    # and what this package writes says what it is
    back = output.read_file(str(ok))
    note = back["_attributes"]["format_note"]
    assert (note.decode() if isinstance(note, bytes) else note) == FORMAT_NOTE
