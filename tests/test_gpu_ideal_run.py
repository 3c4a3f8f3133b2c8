"""SURVEY 8(f) row 2 end to end, as far as this image allows: an ideal case read from init.nc + forcing.nc (the reference's
variable / dimension names; fixture generator tests/golden/make_ideal_files.py) into domain_t, stepped on the device, written
as an output file and a restart file with the reference's naming (driver.f90:94-97), and continued from the restart file."""
import datetime
import os
import sys
import numpy as np
import pytest
from scipy.io import netcdf_file

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_ideal_files as M
from icar_amd import ideal, ideal_io
from icar_amd.domain import domain_t
from icar_amd.grid import grid_t
from icar_amd.options import options_t
from icar_amd.time_step import step, update_dt
from icar_amd.microphysics import mp_init, mp_var_request
from icar_amd.advection import adv_init, adv_var_request
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON, ADVECTION_ORDER
from icar_amd.output import output_t, output_filename, MEMBER
from icar_amd.restart import restart_model

pytestmark = pytest.mark.gpu


def test_ideal_case_from_files_ten_steps_output_and_restart(tmp_path):
    init, forcing = M.write(str(tmp_path))
    opt = options_t()
    opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.ideal = True; opt.parameters.dx = M.DX; opt.parameters.dz_levels = ideal.dz_levels(M.NZ)
    mp_var_request(opt); adv_var_request(opt)
    c = ideal_io.read_ideal_case(init, forcing, opt.parameters.dz_levels, M.DX)
    c["water_vapor"] = (c["water_vapor"] * np.float32(2.5)).astype(np.float32)           # moist enough for the hill to make cloud

    def fresh():
        d = domain_t(grid_t().set_grid_dimensions(M.NX, M.NY, M.NZ, 1, 1), device=0, dx=M.DX)
        d.load_case(c)
        d.exchange_vars = [n for n in ADVECTION_ORDER if opt.vars_to_advect.get(n, 0) > 0]
        mp_init(opt, d); adv_init(d, opt)
        return d
    d = fresh()
    dt = update_dt(d, opt)
    n = step(d, 10.0 * dt, opt)                # ten whole steps (mp_driver.f90's last_model_time is not part of a restart file: a
    assert n in (10, 11)                       # shortened last step would change the next mp_dt in the reference as well)
    names = ["potential_temperature", "water_vapor", "cloud_water", "rain_in_air", "snow_in_air", "cloud_ice", "graupel_in_air",
             "ice_number_concentration", "rain_number_concentration", "precipitation", "u", "v", "w", "pressure"]
    start = datetime.datetime(2020, 12, 1)                                              # Forcing.py:52
    when = start + datetime.timedelta(seconds=d.model_time_seconds)
    out = output_t(image=1); out.add_variables(names)
    fn_out = str(tmp_path / output_filename("icar_out_", 1, when))                      # gen_ideal_test.py:97 output_file = 'icar_out_'
    fn_rst = str(tmp_path / output_filename("icar_rst_", 1, when))
    assert os.path.basename(fn_out).startswith("icar_out_000001_2020-12-01_00-")
    mjd = 59184.0 + d.model_time_seconds / 86400.0
    out.save_file(d, fn_out, 1, mjd); out.save_file(d, fn_rst, 1, mjd)
    with netcdf_file(fn_out, "r", mmap=False) as f:
        qv = f.variables["qv"]
        assert qv.dimensions == ("time", "level", "lat_y", "lon_x") and qv.shape == (1, M.NZ, M.NY, M.NX)
        assert np.array_equal(np.transpose(qv[0], (1, 0, 2)), d.get("water_vapor"))
        assert float(f.variables["qc"][0].max()) > 1e-6, "the hill must have made cloud in ten steps"
    t1 = d.model_time_seconds
    step(d, t1 + 5.0 * dt, opt)
    # a NEW domain continued from the restart record == the uninterrupted run
    d2 = fresh()
    opt.parameters.restart_file = fn_rst; opt.parameters.restart_step_in_file = 1
    restart_model(d2, out, opt)
    d2.model_time_seconds = t1
    step(d2, t1 + 5.0 * dt, opt)
    for nme in names:
        a, b = d.get(MEMBER[nme]), d2.get(MEMBER[nme])
        assert np.array_equal(a, b), nme
    d.close(); d2.close()
