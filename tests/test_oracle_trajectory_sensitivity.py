"""How far can ANY implementation of [mp_simple -> MPDATA] stay from the reference's trajectory without matching its rounding?
The CPU oracle (bit-identical to the compiled reference) is run twice over ten un-resynchronised sub-steps of the ideal hill case,
the second time with every advected value perturbed by at most HALF AN ULP (6e-8 of itself) after each advection -- less than the
1-ulp reciprocals of the device's MPDATA leave.  The two runs agree to rounding after one sub-step and differ in > 5 % of the
cloud-water / rain cells (by percents of the field maximum) after ten: the saturation adjustment and the autoconversion
thresholds amplify rounding.  That is the yardstick for tests/test_gpu_trajectory.py (device vs oracle: the same magnitudes)."""
import numpy as np
from icar_amd import ideal
from util import field_stats


def test_half_ulp_noise_moves_the_trajectory_as_much_as_the_device_differs(oracle):
    nx, ny, nz, nsteps = 64, 48, 40, 10
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.6)).astype(np.float32)
    dt = float(np.float32(min(ideal.cfl_dt(c), 60.0)))
    names = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature"]
    oracle.set_math_mode(0)

    def run(eps):
        s = {n: c[n].copy() for n in names}
        rng = np.random.default_rng(7); out = []
        for _ in range(nsteps):
            rain = np.zeros((ny, nx), np.float32); snow = rain.copy()
            oracle.mp_simple(c["pressure"], s["potential_temperature"], c["exner"], c["density"], s["water_vapor"], s["cloud_water"],
                             s["rain"], s["snow"], rain, snow, dt, c["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
            q = np.stack([s[n] for n in names]).copy()
            oracle.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                          c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
            for m, n in enumerate(names):
                x = q[m].astype(np.float64)
                s[n] = (x + eps * np.abs(x) * rng.uniform(-1, 1, x.shape)).astype(np.float32) if eps else q[m].copy()
            out.append({n: s[n].copy() for n in names})
        return out
    a, b = run(0.0), run(6e-8)
    first = {n: field_stats(b[0][n], a[0][n]) for n in names}
    last = {n: field_stats(b[-1][n], a[-1][n]) for n in names}
    assert max(st["max_abs_over_max"] for st in first.values()) < 2e-7 and max(st["beyond_rtol_frac"] for st in first.values()) == 0.0
    assert float(a[-1]["cloud_water"].max()) > 1e-5
    worst_frac = max(st["beyond_rtol_frac"] for st in last.values()); worst_abs = max(st["max_abs_over_max"] for st in last.values())
    print(f"half-ulp noise, after {nsteps} sub-steps: {worst_frac:.3g} of the cells beyond 1e-5, max |d| / max = {worst_abs:.3g}")
    assert worst_frac > 0.02 and worst_abs > 1e-3, (worst_frac, worst_abs)
