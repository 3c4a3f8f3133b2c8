"""The CPU oracle (oracle/icar_oracle.c) against golden vectors produced by the REFERENCE's own
compiled kernels (tests/golden/make_golden.py, oracle/_ref).  Bit-exact."""
import json
import os
import numpy as np
import pytest
from icar_amd import ideal
from util import bits_equal, nbitdiff

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INPUTS_ADV = ["u", "v", "w", "density", "jacobian", "jacobian_u", "jacobian_v", "jacobian_w", "advection_dz", "dz_levels"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z, json.loads(str(z["params"]))


def summary(a):
    a64 = a.astype(np.float64)
    return dict(sum=float(a64.sum()), min=float(a.min()), max=float(a.max()), sumsq=float((a64 * a64).sum()))


SMALL_CASES = ["adv_upwind_24x20x10", "adv_mpdata_24x20x10", "adv_mpdata_dens_40x36x12", "adv_mpdata_nofct_40x36x12", "adv_mpdata_order1_40x36x12",
               "adv_mpdata_rough_40x36x12", "adv_mpdata_rough_dens_order3_40x36x12"]      # rough: white-noise winds (round 6)
CONFIG1_CASES = ["adv_mpdata_100x100x30", "adv_upwind_100x100x30"]
MP_SIMPLE_CASES = ["mp_simple_40x36x20", "mp_simple_snow_30x20x30"]        # (tests/golden_pin.py runs the same lists inside a -m gpu session)


@pytest.mark.parametrize("name", SMALL_CASES)
def test_advection_golden_small(oracle, name):
    z, p = load(name)
    c = {n: np.ascontiguousarray(z["in_" + n]) for n in INPUTS_ADV}
    q = np.stack([z["in_" + n] for n in p["vars"]]).copy()
    oracle.advect(p["scheme"], q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"],
                  c["jacobian_w"], c["advection_dz"], c["dz_levels"], 1000.0, float(z["dt"]), advect_density=p["dens"],
                  mpdata_order=p["order"], fct=p["fct"], nsteps=p["nsteps"])
    assert bits_equal(q, z["q"]), f"{nbitdiff(q, z['q'])} values differ from the reference"


@pytest.mark.parametrize("name", CONFIG1_CASES)
def test_advection_golden_config1_grid(oracle, name):
    """BASELINE config[0] grid, 10 steps; inputs regenerated from IEEE-exact arithmetic."""
    z, p = load(name)
    c = ideal.make_case(p["nx"], p["ny"], p["nz"], hill_height=p["hill"], noise=0.01, n_hydro=1, exact=True)
    fp = sum(float(c[n].astype(np.float64).sum()) for n in p["vars"])
    assert fp == float(z["input_sum"]), "icar_amd.ideal(exact=True) no longer reproduces the fixture inputs"
    dt = ideal.cfl_dt(c)
    assert dt == float(z["dt"])
    q = np.stack([c[n] for n in p["vars"]]).copy()
    oracle.advect(p["scheme"], q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"],
                  c["jacobian_w"], c["advection_dz"], c["dz_levels"], float(c["dx"]), dt, advect_density=p["dens"],
                  mpdata_order=p["order"], fct=p["fct"], nsteps=p["nsteps"])
    want = json.loads(str(z["summary"]))
    for m, n in enumerate(p["vars"]):
        assert summary(q[m]) == want[n]
    assert bits_equal(q[:, 50], z["plane_j50"])


@pytest.mark.parametrize("name", MP_SIMPLE_CASES)
def test_mp_simple_golden(oracle, name):
    z, p = load(name)
    nx, ny, nz = p["nx"], p["ny"], p["nz"]
    s = {k: np.ascontiguousarray(z["in_" + k]).copy() for k in ["pressure", "potential_temperature", "exner", "density",
                                                                 "water_vapor", "cloud_water", "rain", "snow", "dz_mass"]}
    rain = np.zeros((ny, nx), np.float32); snow = np.zeros((ny, nx), np.float32)
    oracle.set_math_mode(0)
    for _ in range(p["nsteps"]):
        err = oracle.mp_simple(s["pressure"], s["potential_temperature"], s["exner"], s["density"], s["water_vapor"],
                               s["cloud_water"], s["rain"], s["snow"], rain, snow, p["dt"], s["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
        assert err == 0
        s["potential_temperature"] -= np.float32(p["cool"])
    for k in ["potential_temperature", "water_vapor", "cloud_water", "rain", "snow"]:
        assert bits_equal(s[k], z[k]), f"{k}: {nbitdiff(s[k], z[k])} values differ from the reference"
    assert bits_equal(rain, z["rain_acc"]) and bits_equal(snow, z["snow_acc"])
    if "snow" in name:
        assert z["snow"].max() > 0 and z["snow_acc"].max() >= 0
    assert z["rain_acc"].max() > 0 and z["cloud_water"].max() > 0


def test_survey_sanity_values(oracle):
    """24x10x20 tile, u=5 v=3 w=0, 10 MPDATA then 10 upwind steps; values
    printed by the compiled reference."""
    nx, nz, ny = 24, 10, 20
    i = np.arange(1, nx + 1, dtype=np.float32)[None, None, :]; j = np.arange(1, ny + 1, dtype=np.float32)[:, None, None]
    qv = (np.float32(0.001) + np.float32(0.004) * np.exp(-((i - np.float32(12.)) ** 2 + (j - np.float32(10.)) ** 2) / np.float32(9.0))).astype(np.float32)
    q = np.ascontiguousarray(np.broadcast_to(qv, (ny, nz, nx))[None].copy())
    u = np.full((ny, nz, nx + 1), 5, np.float32); v = np.full((ny + 1, nz, nx), 3, np.float32); w = np.zeros((ny, nz, nx), np.float32)
    one = np.ones((ny, nz, nx), np.float32); ju = np.ones((ny, nz, nx + 1), np.float32); jv = np.ones((ny + 1, nz, nx), np.float32)
    dz = np.full((ny, nz, nx), 200, np.float32); dzl = np.full(nz, 200, np.float32)
    oracle.advect(2, q, u, v, w, one, one, ju, jv, one, dz, dzl, 1000., 20., nsteps=10)
    assert abs(q.astype(np.float64).sum() - 5.93093799) < 2e-6 and abs(float(q.max()) - 4.57896292e-3) < 1e-9
    oracle.advect(1, q, u, v, w, one, one, ju, jv, one, dz, dzl, 1000., 20., nsteps=10)
    assert abs(q.astype(np.float64).sum() - 5.93058809) < 2e-6 and abs(float(q.max()) - 4.18629264e-3) < 1e-9
