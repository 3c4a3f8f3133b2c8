"""tests/golden/make_trajectory_sensitivity.py -> tests/golden/trajectory_sensitivity.json  (CPU only, ~2 min; rerun when the case of
tests/test_gpu_trajectory.py::test_trajectory_ten_unsynchronised_substeps changes).

The yardstick for that test: the CPU oracle (bit-identical to the compiled reference) run over the same ten un-resynchronised
sub-steps of the same 128 x 96 x 40 case, against ITSELF with every advected value perturbed by at most eps of itself after each
advection -- eps = 6e-8 (half an ulp) and 1.2e-7 (one ulp; the fused MPDATA kernel's 1-ulp reciprocals leave <= 2e-7 of the local
scale per step, profiles/r06_parity.json).  For each scheme, eps and recorded sub-step: the maximum over 8 noise seeds of the worst
field's fraction of cells beyond 1e-5 and max |d| / max.  The device test asserts <= 2 x the one-ulp figures: a bound that comes from
the reference's own sensitivity, not from what some kernel happened to produce."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from icar_amd import ideal
from icar_amd.options import options_t
from oracle import orc
from util import field_stats

ADV_ORDER = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]
RECORD = (1, 2, 3, 5, 10)


def run(scheme, c, dt, eps, seed, nsteps=10):
    ny, nz, nx = c["water_vapor"].shape
    names = ADV_ORDER if scheme == "thompson" else ADV_ORDER[:5]
    s = {n: c[n].copy() for n in names}
    rng = np.random.default_rng(seed); out = {}
    for it in range(nsteps):
        if scheme == "thompson":
            z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
            orc.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                         s["rain_number"], s["potential_temperature"], c["exner"], c["pressure"], c["dz_mass"], dt, *z,
                         1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
        else:
            rain = np.zeros((ny, nx), np.float32); snow = rain.copy()
            orc.mp_simple(c["pressure"], s["potential_temperature"], c["exner"], c["density"], s["water_vapor"], s["cloud_water"],
                          s["rain"], s["snow"], rain, snow, dt, c["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
        q = np.stack([s[n] for n in names]).copy()
        orc.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                   c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
        for m, n in enumerate(names):
            x = q[m].astype(np.float64)
            s[n] = (x + eps * np.abs(x) * rng.uniform(-1, 1, x.shape)).astype(np.float32) if eps else q[m].copy()
        if it + 1 in RECORD:
            out[it + 1] = {n: s[n].copy() for n in names}
    return out


def main():
    orc.build()
    orc.thompson_init(*options_t().mp_options.as_arrays()); orc.set_math_mode(0)
    nx, ny, nz = 128, 96, 40
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.6)).astype(np.float32)
    dt = float(np.float32(min(ideal.cfl_dt(c), 60.0)))
    res = {"case": "ideal.make_case(128, 96, 40, hill_height=1000, noise=0.01, n_hydro=1), water_vapor x 1.6", "dt": dt, "seeds": 8, "schemes": {}}
    for scheme in ("thompson", "simple"):
        base = run(scheme, c, dt, 0.0, 0)
        per_eps = {}
        for eps in (6e-8, 1.2e-7):
            worst = {str(k): {"beyond_rtol_frac": 0.0, "max_abs_over_max": 0.0} for k in RECORD}
            for seed in range(8):
                got = run(scheme, c, dt, eps, 100 + seed)
                for k in RECORD:
                    st = [field_stats(got[k][n], base[k][n], 1e-5) for n in base[k]]
                    for key in ("beyond_rtol_frac", "max_abs_over_max"):
                        worst[str(k)][key] = max(worst[str(k)][key], max(x[key] for x in st))
            per_eps[f"{eps:g}"] = worst
            print(scheme, eps, json.dumps(worst), flush=True)
        res["schemes"][scheme] = per_eps
    json.dump(res, open(os.path.join(HERE, "trajectory_sensitivity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
