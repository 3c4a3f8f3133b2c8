"""WSM6 (src/physics/mp_wsm6.f90): the CPU restatement oracle/wsm6_oracle.c against the UNMODIFIED reference
module compiled into oracle/_ref -- bit for bit:
  * the 60 constants wsm6init derives (rgmma's 10000-term products, the x**y of the slope limits ...);
  * whole tiles over several calls of wsm6 as mp_driver.f90:518-550 makes them: warm rain, graupel / snow / cloud ice aloft with
    melting below, snow and graupel at the surface, a surface that crosses 0 C, dt > 180 s (two minor loops)."""
import numpy as np
import pytest
from icar_amd import ideal

ref = pytest.importorskip("oracle.ref")
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def test_wsm6init_constants(oracle):
    want = ref.wsm6_init()
    got = oracle.wsm6_init()
    for n, w in zip(oracle.WSM6_CONSTS, want):
        assert np.float32(got[n]).view(np.int32) == np.float32(w).view(np.int32), (n, float(got[n]), float(w))
    assert abs(float(got["g4pbr"]) - 17.837825) < 0.1 and abs(float(got["pi"]) - np.pi) < 1e-6 and float(got["pidn0g"]) > 0


CASES = {"warm_rain": dict(nx=33, ny=21, nz=25, steps=8, dt=45.0, moist=1.8, cool0=0.0, cool=1.0, seed=3),
         "two_minor_loops": dict(nx=36, ny=19, nz=30, steps=5, dt=200.0, moist=2.2, cool0=0.0, cool=7.0, seed=9),
         "snow_at_surface": dict(nx=33, ny=21, nz=25, steps=10, dt=60.0, moist=1.3, cool0=28.0, cool=0.5, seed=3),
         "cold_long_step": dict(nx=30, ny=16, nz=30, steps=6, dt=200.0, moist=1.2, cool0=40.0, cool=0.2, seed=5),
         "surface_crosses_0C": dict(nx=24, ny=14, nz=20, steps=12, dt=90.0, moist=1.5, cool0=22.0, cool=1.0, seed=8),
         "mixed_phase": dict(nx=28, ny=15, nz=32, steps=14, dt=75.0, moist=2.0, cool0=8.0, cool=1.5, seed=11)}
KEYS = ["potential_temperature", "water_vapor", "cloud_water", "rain", "cloud_ice", "snow", "graupel"]


def wsm6_state(c, k):
    """the seven prognostic fields of a case, moistened; hydrometeors seeded so that every class is present from the first call"""
    rng = np.random.default_rng(k["seed"])
    A = {"potential_temperature": c["potential_temperature"].copy(), "water_vapor": (c["water_vapor"] * np.float32(k["moist"])).astype(np.float32),
         "cloud_water": c["cloud_water"].copy(), "rain": c["rain"].copy()}
    shape = c["water_vapor"].shape
    for n, amp in (("cloud_ice", 2e-5), ("snow", 2e-4), ("graupel", 1e-4)):
        f = (amp * rng.random(shape) ** 3).astype(np.float32)
        f[rng.random(shape) < 0.4] = 0.0
        A[n] = f
    return A


@pytest.mark.parametrize("case", list(CASES))
def test_wsm6_tiles_bit_exact(oracle, case):
    k = CASES[case]
    nx, ny, nz, dt = k["nx"], k["ny"], k["nz"], k["dt"]
    oracle.set_math_mode(0)
    _, args = ref.wsm3_init(); ref.wsm6_init(); oracle.wsm6_init()       # the 18 scalars mp_driver passes are the same for both schemes
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.03, seed=k["seed"], n_hydro=1, cool=k["cool0"])
    A = wsm6_state(c, k); B = {n: v.copy() for n, v in A.items()}
    z2 = lambda: np.zeros((ny, nx), np.float32)
    ra = dict(rain=z2(), rainncv=z2(), sr=z2(), snow=z2(), graupel=z2()); rb = dict(rain=z2(), sr=z2(), snow=z2(), graupel=z2())
    a18 = args.copy(); a18[0] = dt
    for s in range(k["steps"]):
        ref.wsm6(A["potential_temperature"], A["water_vapor"], A["cloud_water"], A["rain"], A["cloud_ice"], A["snow"], A["graupel"], c["density"],
                 c["exner"], c["pressure"], c["dz_mass"], dt, ra["rain"], ra["rainncv"], ra["sr"], ra["snow"], ra["graupel"], 2, nx - 1, 2, ny - 1, 1, nz)
        assert oracle.wsm6(B["potential_temperature"], B["water_vapor"], B["cloud_water"], B["rain"], B["cloud_ice"], B["snow"], B["graupel"],
                           c["density"], c["exner"], c["pressure"], c["dz_mass"], a18, rb["rain"], rb["sr"], rb["snow"], rb["graupel"],
                           2, nx - 1, 2, ny - 1, 1, nz) == 0
        for n in KEYS:
            assert np.array_equal(A[n].view(np.int32), B[n].view(np.int32)), f"call {s}: {n}: {np.count_nonzero(A[n] != B[n])} cells differ"
        A["potential_temperature"] -= np.float32(k["cool"]); B["potential_temperature"] -= np.float32(k["cool"])
    for n in ("rain", "sr", "snow", "graupel"):
        assert np.array_equal(ra[n].view(np.int32), rb[n].view(np.int32)), n
    assert ra["rainncv"].max() == 0                                               # the reference's rainncv lines are commented out
    assert ra["rain"].max() > 0.1 and A["cloud_water"].max() > 1e-5
    assert np.array_equal(A["water_vapor"][0], (c["water_vapor"] * np.float32(k["moist"])).astype(np.float32)[0])   # ring untouched
    if case in ("snow_at_surface", "cold_long_step"):
        assert ra["snow"].max() > 0.05 and ra["graupel"].max() > 0
    if case in ("warm_rain", "mixed_phase"):
        assert A["rain"].max() > 1e-5
