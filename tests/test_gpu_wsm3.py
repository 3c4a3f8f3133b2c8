"""WSM3 on the device (icar_amd/csrc/mp_wsm3.hip) through mp()'s dispatch vs the CPU oracle (oracle/wsm3_oracle.c, itself
pinned bit-for-bit to the compiled mp_wsm3.f90 in tests/test_oracle_wsm3.py):
  * oracle math mode 1 (exp / log / x**y = FP64 function rounded once, as the device evaluates them): BIT-EXACT;
  * oracle math mode 0 (libm, = the reference): rtol 1e-5 on all but a small share of cells (a 1-ulp change of a
    transcendental can flip one of the scheme's threshold tests), precipitation within 1e-4 relative."""
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.microphysics import mp, mp_init, mp_var_request
from icar_amd.constants import kMP_WSM3
from util import single_image_domain

pytestmark = pytest.mark.gpu
CASES = {"warm_rain": dict(nx=70, ny=21, nz=25, steps=6, dt=45.0, moist=1.8, cool0=0.0, cool=1.0, seed=3),
         "snow_at_surface": dict(nx=66, ny=20, nz=30, steps=8, dt=60.0, moist=1.3, cool0=28.0, cool=0.5, seed=4),
         "two_minor_loops_40_levels": dict(nx=40, ny=17, nz=40, steps=4, dt=200.0, moist=1.5, cool0=22.0, cool=1.0, seed=8),
         # 64 levels (the build's maximum): the fall needs 65 lanes, so it runs one thread per column (k_wsm3_fall)
         "serial_fall_64_levels": dict(nx=34, ny=12, nz=64, steps=3, dt=90.0, moist=1.6, cool0=10.0, cool=1.0, seed=5, uniform_dz=150.0)}
ARGS18 = np.array([0, 9.81, 1012.0, 4 * np.float32(461.6), 287.058, 461.5, 273.15, np.float32(461.5) / np.float32(287.058) - np.float32(1),
                   np.float32(287.058) / np.float32(461.5), 1e-15, 2.85e6, 2.5e6, 3.5e5, 1.28, 1000.0, 4190.0, 2106.0, 610.78], np.float32)


def run(oracle, k, mode, split=False):
    nx, ny, nz, dt = k["nx"], k["ny"], k["nz"], k["dt"]
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.03, seed=k["seed"], n_hydro=1, cool=k["cool0"], uniform_dz=k.get("uniform_dz"))
    c["water_vapor"] = (c["water_vapor"] * np.float32(k["moist"])).astype(np.float32)
    c["w_real"] = (c["w"] + 0.3 * np.random.default_rng(k["seed"]).standard_normal(c["w"].shape)).astype(np.float32)
    keys = ["potential_temperature", "water_vapor", "cloud_water", "rain"]
    B = {n: c[n].copy() for n in keys}
    d = single_image_domain(c); d.set("w_real", c["w_real"])
    opt = options_t(); opt.physics.microphysics = kMP_WSM3
    mp_var_request(opt); mp_init(opt, d)
    assert opt.vars_to_advect.get("rain_in_air", 0) > 0 and opt.vars_to_advect.get("snow_in_air", 0) == 0
    z2 = lambda: np.zeros((ny, nx), np.float32)
    acc_r = np.zeros((ny, nx), np.float64); acc_s = np.zeros((ny, nx), np.float64)
    a18 = ARGS18.copy(); a18[0] = dt
    oracle.set_math_mode(mode)
    try:
        oracle.wsm3_init()
        for s in range(k["steps"]):
            rb = [z2() for _ in range(5)]
            assert oracle.wsm3(B["potential_temperature"], B["water_vapor"], B["cloud_water"], B["rain"], c["w_real"], c["density"], c["exner"],
                               c["pressure"], c["dz_mass"], a18, *rb, 2, nx - 1, 2, ny - 1, 1, nz) == 0
            acc_r += rb[0]; acc_s += rb[2]
            B["potential_temperature"] -= np.float32(k["cool"])
            if split == "two_streams":
                from icar_amd.time_step import mp_and_halo
                mp_and_halo(d, opt, dt)      # strips on the main stream, interior on the second one
            elif split:
                mp(d, opt, dt, halo=1); mp(d, opt, dt, subset=1)      # strips + interior == whole tile
            else:
                mp(d, opt, dt)
            d.model_time_seconds += dt
            d.set("potential_temperature", d.get("potential_temperature") - np.float32(k["cool"]))
    finally:
        oracle.set_math_mode(0)
    got = {"potential_temperature": d.get("potential_temperature"), "water_vapor": d.get("water_vapor"), "cloud_water": d.get("cloud_water_mass"),
           "rain": d.get("rain_mass")}
    pa, sa = d.get("accumulated_precipitation"), d.get("accumulated_snowfall")
    d.close()
    return got, B, pa, sa, acc_r, acc_s


@pytest.mark.parametrize("case", list(CASES))
def test_wsm3_bit_exact_vs_reference_math(oracle, case):
    got, want, pa, sa, acc_r, acc_s = run(oracle, CASES[case], mode=0, split={"warm_rain": True, "snow_at_surface": "two_streams"}.get(case, False))
    for n in want:
        assert np.array_equal(got[n].view(np.int32), want[n].view(np.int32)), f"{n}: {(got[n] != want[n]).sum()} cells differ"
    assert np.array_equal(pa, acc_r) and np.array_equal(sa, acc_s) and acc_r.max() > (0.5 if "serial" not in case else 0.0)
    if case not in ("warm_rain", "serial_fall_64_levels"):
        assert acc_s.max() > (0.1 if case == "snow_at_surface" else 0.0)




def test_wsm3_full_size_every_column_bit_exact(oracle):
    """512 x 512 x 40: every cell of two calls, device vs the checker's slab restatement (oracle/wsm3_oracle.c), bit for bit"""
    k = dict(nx=512, ny=512, nz=40, steps=2, dt=60.0, moist=1.5, cool0=15.0, cool=1.0, seed=21)
    got, want, pa, sa, acc_r, acc_s = run(oracle, k, mode=0)
    for n in want:
        assert np.array_equal(got[n].view(np.int32), want[n].view(np.int32)), f"{n}: {(got[n] != want[n]).sum()} cells differ"
    assert np.array_equal(pa, acc_r) and np.array_equal(sa, acc_s) and acc_r.max() > 0
