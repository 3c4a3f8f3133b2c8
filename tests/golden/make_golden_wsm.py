#!/usr/bin/env python
"""Golden vectors of WSM3 / WSM6 under tests/golden/ made by RUNNING THE REFERENCE'S OWN KERNELS (oracle/_ref/libicar_ref.so =
/root/reference/src/physics/mp_wsm3.f90 / mp_wsm6.f90 compiled unmodified, oracle/build_ref.sh): the constants wsm3init / wsm6init
derive, the 18 scalars mp_driver.f90 passes, and the state + surface accumulators after several calls on small tiles.  Inputs are
regenerated from icar_amd.ideal with the recorded parameters (a fingerprint detects drift); the expected outputs are stored.
Only runs where /root/reference is present; tests/test_oracle_wsm_golden.py pins the CPU restatements to these files everywhere
(the GPU box has no /root/reference)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {"wsm3_warm_two_loops_26x14x24": dict(scheme=3, nx=26, ny=14, nz=24, steps=5, dt=200.0, moist=2.0, cool0=0.0, cool=5.0, seed=31),
         "wsm3_snow_crossing_0C_24x12x30": dict(scheme=3, nx=24, ny=12, nz=30, steps=9, dt=75.0, moist=1.4, cool0=22.0, cool=1.0, seed=32),
         "wsm6_mixed_phase_24x12x30": dict(scheme=6, nx=24, ny=12, nz=30, steps=10, dt=75.0, moist=2.0, cool0=8.0, cool=1.5, seed=33),
         "wsm6_cold_two_loops_22x10x28": dict(scheme=6, nx=22, ny=10, nz=28, steps=5, dt=200.0, moist=1.3, cool0=35.0, cool=0.3, seed=34)}
K3 = ["potential_temperature", "water_vapor", "cloud_water", "rain"]
K6 = K3 + ["cloud_ice", "snow", "graupel"]


def initial_state(c, p):
    """the prognostic fields of a case (shared with the test): moistened; WSM6's ice classes seeded so that every class is present"""
    A = {"potential_temperature": c["potential_temperature"].copy(), "water_vapor": (c["water_vapor"] * np.float32(p["moist"])).astype(np.float32),
         "cloud_water": c["cloud_water"].copy(), "rain": c["rain"].copy()}
    if p["scheme"] == 6:
        rng = np.random.default_rng(p["seed"])
        for n, amp in (("cloud_ice", 2e-5), ("snow", 2e-4), ("graupel", 1e-4)):
            f = (amp * rng.random(c["water_vapor"].shape) ** 3).astype(np.float32)
            f[rng.random(f.shape) < 0.4] = 0.0
            A[n] = f
    return A


def make_inputs(p):
    from icar_amd import ideal
    c = ideal.make_case(p["nx"], p["ny"], p["nz"], hill_height=800.0, noise=0.03, seed=p["seed"], n_hydro=1, cool=p["cool0"])
    w = (c["w"] + 0.3 * np.random.default_rng(p["seed"]).standard_normal(c["w"].shape)).astype(np.float32)
    return c, w, initial_state(c, p)


def fingerprint(c, w, A):
    return float(sum(float(np.asarray(x, np.float64).sum()) for x in list(A.values()) + [w, c["density"], c["exner"], c["pressure"], c["dz_mass"]]))


def run_case(name):
    from oracle import ref
    p = CASES[name]
    nx, ny, nz, dt = p["nx"], p["ny"], p["nz"], p["dt"]
    c, w, A = make_inputs(p)
    out = {"input_fingerprint": np.float64(fingerprint(c, w, A))}
    c3, args = ref.wsm3_init()
    out["args18"] = args
    z2 = lambda: np.zeros((ny, nx), np.float32)
    if p["scheme"] == 3:
        out["consts"] = c3
        acc = [z2() for _ in range(5)]                                  # rain, rainncv, snow, snowncv, sr
        for _ in range(p["steps"]):
            ref.wsm3(A["potential_temperature"], A["water_vapor"], A["cloud_water"], A["rain"], w, c["density"], c["exner"], c["pressure"],
                     c["dz_mass"], dt, *acc, 2, nx - 1, 2, ny - 1, 1, nz)
            A["potential_temperature"] -= np.float32(p["cool"])
        for n, a in zip(("rain", "rainncv", "snow", "snowncv", "sr"), acc):
            out["acc_" + n] = a
        keys = K3
    else:
        out["consts"] = ref.wsm6_init()
        acc = dict(rain=z2(), rainncv=z2(), sr=z2(), snow=z2(), graupel=z2())
        for _ in range(p["steps"]):
            ref.wsm6(A["potential_temperature"], A["water_vapor"], A["cloud_water"], A["rain"], A["cloud_ice"], A["snow"], A["graupel"], c["density"],
                     c["exner"], c["pressure"], c["dz_mass"], dt, acc["rain"], acc["rainncv"], acc["sr"], acc["snow"], acc["graupel"], 2, nx - 1, 2, ny - 1, 1, nz)
            A["potential_temperature"] -= np.float32(p["cool"])
        for n, a in acc.items():
            out["acc_" + n] = a
        keys = K6
    for n in keys:
        out[n] = A[n]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=np.array(json.dumps(p)), **out)
    print("wrote", name, {n: float(A[n].max()) for n in keys[2:]}, "rain", float(out["acc_rain"].max()), "snow", float(out["acc_snow"].max()))


if __name__ == "__main__":
    for n in (sys.argv[1:] or CASES):
        run_case(n)
