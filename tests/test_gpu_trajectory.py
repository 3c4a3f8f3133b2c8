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
from icar_amd.capi import lib, check
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON, kMP_SB04
from util import single_image_domain, field_stats, parity_record, local_rel_err, assert_fields_close, bits_equal, nbitdiff, SCALARS, MEMBER

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


# How far the device may be from the oracle after k un-resynchronised sub-steps is NOT read off the device (round 5 did that and the
# bound followed the kernel).  It is the reference's own sensitivity: tests/golden/trajectory_sensitivity.json
# (tests/golden/make_trajectory_sensitivity.py, CPU only) holds, for this very case, the CPU oracle against itself with every advected
# value perturbed by at most ONE ULP after each advection (8 noise seeds, worst field, worst seed) -- the fused MPDATA kernel's 1-ulp
# reciprocals leave <= 2e-7 of the local scale per step, about that much.  After ONE sub-step that is 1.4e-7 of the field maximum and
# no cell beyond 1e-5; after two, the microphysics' threshold tests have amplified it to 14-16 % of the cells and 3-30 % of the field
# maximum (one cell of cloud water on the other side of the autoconversion threshold is 2.7 % of the maximum -- half an ulp does that
# too).  The device has to stay within TRAJ_FACTOR x the one-ulp figures at every recorded sub-step.
TRAJ_FACTOR = 1.25


def _sensitivity_bounds(scheme):
    import json, os
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "trajectory_sensitivity.json")))
    y = z["schemes"][scheme]["1.2e-07"]
    return {int(k): dict(beyond=TRAJ_FACTOR * v["beyond_rtol_frac"], absmax=TRAJ_FACTOR * v["max_abs_over_max"]) for k, v in y.items()}, z


FIRST_STEP_BOUNDS = dict(beyond=1.3e-5, absmax=6e-7)


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
        if it in (0, 1, 2, 4, nsteps - 1):
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
    bounds, z = _sensitivity_bounds(scheme)
    assert abs(z["dt"] - dt) < 1e-6 * dt, "tests/golden/trajectory_sensitivity.json belongs to another case: rerun make_trajectory_sensitivity.py"
    assert worst[1]["beyond_rtol_frac"] <= FIRST_STEP_BOUNDS["beyond"] and worst[1]["max_abs_over_max"] <= FIRST_STEP_BOUNDS["absmax"], (scheme, worst[1])
    for it, w in worst.items():
        if it == 1: continue
        bb = bounds[it]
        assert w["beyond_rtol_frac"] <= bb["beyond"] and w["max_abs_over_max"] <= bb["absmax"], (scheme, it, w, bb)
    assert rel_p <= 1e-3, rel_p
    d.close()


@pytest.mark.parametrize("scheme", ["thompson", "simple"])
def test_trajectory_exact_mode_bit_identical_over_ten_substeps(th_oracle, oracle, scheme):
    """The same ten unsynchronised [microphysics -> MPDATA] sub-steps with icar_hip_mpdata_exact(ctx, 1): the device trajectory
    is BIT-IDENTICAL to the CPU oracle's after every sub-step -- every cell of every advected scalar and the accumulated
    precipitation.  With the advection in the reference's operation order nothing on the path rounds differently from the CPU
    reference, so the drift test_trajectory_ten_unsynchronised_substeps measures is the fused kernel's 1-ulp reciprocals
    amplified by the case, and nothing else."""
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
    check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
    mp_init(opt, d); adv_init(d, opt)
    s = {n: c[n].copy() for n in names}
    acc = np.zeros((ny, nx), np.float64)
    orc = th_oracle if scheme == "thompson" else oracle
    orc.set_math_mode(0)
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
        for n in names:
            got = d.get(MEMBER[n])
            assert bits_equal(got, s[n]), f"{scheme} sub-step {it + 1}, {n}: {nbitdiff(got, s[n])} of {got.size} cells differ"
    assert acc.max() > 0 and float(s["cloud_water"].max()) > 1e-6, "the case must have active microphysics"
    assert np.array_equal(d.get("accumulated_precipitation"), acc)
    parity_record("trajectory", f"{scheme}/128x96x40/exact_mode_{nsteps}_substeps", {n: {"bitdiff_cells": 0, "cells": int(s[n].size)} for n in names})
    d.close()


def test_metric_grid_512x512x40_five_steps_exact_mode_bit_identical(th_oracle, oracle):
    """BASELINE.json's metric configuration as bench.py builds it (512 x 512 x 40, hill 1000 m, 1 % noise, vapour x 1.4, MPDATA order 2
    + FCT of the 9 scalars + Thompson), five steps of icar_hip_step_n (update_dt -> diagnostic_update -> Thompson strips + interior
    -> MPDATA) with icar_hip_mpdata_exact(ctx, 1), against the same five steps of the CPU oracle's operators: all 9 x 10.5 M cells
    and the accumulated precipitation bit for bit."""
    from icar_amd.time_step import step_n
    import os
    nx, ny, nz, nsteps = 512, 512, 40, int(os.environ.get("ICAR_METRIC_STEPS", "5"))      # (run once per round with 40: profiles/r04_parity.json)
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, seed=1234, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.4)).astype(np.float32)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"]); opt.parameters.ideal = True
    mp_var_request(opt)
    d = single_image_domain(c)
    check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
    mp_init(opt, d); adv_init(d, opt)
    d.set("dzdx", np.zeros(c["u"].shape, np.float32)); d.set("dzdy", np.zeros(c["v"].shape, np.float32))
    dt_dev = step_n(d, nsteps, opt, diagnostics=True)
    f32 = np.float32
    dt = min(float(f32(0.9) / f32(oracle.max_courant(c["u"], c["v"], c["w"], c["dz_levels"], float(c["dx"])))), 120.0)
    assert dt_dev == dt
    s = {n: c[n].copy() for n in ADV_ORDER}
    acc = np.zeros((ny, nx), np.float64)
    th_oracle.set_math_mode(0)
    zero_x = np.zeros(c["u"].shape, np.float32); zero_y = np.zeros(c["v"].shape, np.float32)
    for it in range(nsteps):
        diag = oracle.diagnostic_update(c["pressure"], s["potential_temperature"], c["u"], c["v"], c["w"], zero_x, zero_y, c["jacobian"])
        z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
        th_oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                           s["rain_number"], s["potential_temperature"], diag["exner"], c["pressure"], c["dz_mass"], dt, *z,
                           1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
        acc += z[0]
        _advect_oracle(th_oracle, s, c, dt, ADV_ORDER)
    for n in ADV_ORDER:
        got = d.get(MEMBER[n])
        assert bits_equal(got, s[n]), f"{n}: {nbitdiff(got, s[n])} of {got.size} cells differ after {nsteps} steps"
        assert np.isfinite(got).all()
    assert acc.max() > 0 and np.array_equal(d.get("accumulated_precipitation"), acc)
    parity_record("trajectory", f"metric_grid/512x512x40/exact_mode_{nsteps}_steps", {n: {"bitdiff_cells": 0, "cells": int(s[n].size)} for n in ADV_ORDER})
    d.close()


def test_whole_step_loop_exact_mode_equals_cpu_chain(th_oracle, oracle):
    """icar_hip_step -- the library's own loop of time_step.f90:440-551: update_dt (CFL maximum on the device, prefetched beside the
    advection) -> diagnostic_update -> Thompson (strips + interior on two streams) -> halo self-exchange -> MPDATA -> apply_forcing
    of qv, theta, u, v, w, p -> enforce_limits in the last two sub-steps, the last sub-step shortened to the end time -- with
    icar_hip_mpdata_exact(ctx, 1), against the SAME loop assembled on the CPU from the oracle's operators.  The winds and the
    pressure are forced, so every sub-step has its own dt, exner and Courant winds.  Sub-step count, every dt, every prognostic
    field, exner / density and the accumulated precipitation: bit for bit."""
    from icar_amd.time_step import step
    nx, ny, nz = 96, 64, 20
    c = ideal.make_case(nx, ny, nz, hill_height=900.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.35)).astype(np.float32)
    rng = np.random.default_rng(5)
    c["dzdx"] = (0.05 * rng.standard_normal(c["u"].shape)).astype(np.float32)
    c["dzdy"] = (0.05 * rng.standard_normal(c["v"].shape)).astype(np.float32)
    dq = {"water_vapor": 1e-8, "potential_temperature": 1e-4, "u": 5e-4, "v": -5e-4, "pressure": 1e-3, "w": 2e-6}
    dq = {k: (sc * rng.standard_normal(c[k].shape)).astype(np.float32) for k, sc in dq.items()}
    forced = [("water_vapor", True), ("potential_temperature", True), ("u", False), ("v", False), ("pressure", False), ("w", False)]
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"]); opt.parameters.ideal = True
    mp_var_request(opt)
    d = single_image_domain(c)
    check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
    mp_init(opt, d); adv_init(d, opt)
    for k, a in dq.items():
        d.set_dqdt(k, a)
    f32 = np.float32
    dt0 = min(float(f32(0.9) / f32(oracle.max_courant(c["u"], c["v"], c["w"], c["dz_levels"], float(c["dx"])))), 120.0)
    end = 6.4 * dt0                                           # ~7 sub-steps, the last one shortened
    n_dev = step(d, end, opt, forced=forced, diagnostics=True)
    # ---- the same loop on the CPU ----
    s = {k: c[k].copy() for k in ADV_ORDER + ["u", "v", "w", "pressure"]}
    acc = np.zeros((ny, nx), np.float64)
    th_oracle.set_math_mode(0); oracle.set_math_mode(0)
    t, n_cpu, dts, t_mp = 0.0, 0, [], None
    while t < end:                                                                                      # :462
        dt = min(float(f32(0.9) / f32(oracle.max_courant(s["u"], s["v"], s["w"], c["dz_levels"], float(c["dx"])))), 120.0)   # :465, :417
        if t + dt > end: dt = end - t                                                                   # :469-471
        enforce = (end - t) < dt * 2
        dt4 = float(f32(dt))
        diag = oracle.diagnostic_update(s["pressure"], s["potential_temperature"], s["u"], s["v"], s["w"], c["dzdx"], c["dzdy"], c["jacobian"])   # :474
        z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
        # mp_driver.f90:698-713: the microphysics integrates over the time since ITS last call (= the previous sub-step's dt), the
        # first call over this sub-step's
        mp_dt = dt4 if t_mp is None else float(f32(t - t_mp)); t_mp = t
        th_oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                           s["rain_number"], s["potential_temperature"], diag["exner"], s["pressure"], c["dz_mass"], mp_dt, *z,
                           1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)                             # :512-523
        acc += z[0]
        q = np.stack([s[n] for n in ADV_ORDER]).copy()
        oracle.advect(2, q, s["u"], s["v"], s["w"], diag["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                      c["advection_dz"], c["dz_levels"], float(c["dx"]), dt4)                            # :529
        for m, n in enumerate(ADV_ORDER): s[n] = q[m].copy()
        for n, fb in forced:                                                                            # :534
            oracle.apply_forcing(s[n], dq[n], dt, int(fb), 1, 1, 1, 1)
        if enforce:
            for n in ADV_ORDER: oracle.enforce_limits(s[n])
        t += dt; n_cpu += 1; dts.append(dt)
    assert n_dev == n_cpu and n_cpu >= 6, (n_dev, n_cpu)
    assert all(np.isfinite(a).all() for a in s.values()) and float(s["potential_temperature"].max()) < 600.0, "the case must stay physical"
    assert abs(d.model_time_seconds - t) == 0.0 or abs(d.model_time_seconds - end) < 1e-9
    dev_name = dict(MEMBER); dev_name.update({"u": "u", "v": "v", "w": "w", "pressure": "pressure"})
    for n in ADV_ORDER + ["u", "v", "w", "pressure"]:
        got = d.get(dev_name[n])
        assert bits_equal(got, s[n]), f"{n}: {nbitdiff(got, s[n])} of {got.size} cells differ after {n_cpu} sub-steps"
    assert bits_equal(d.get("exner"), diag["exner"]) and bits_equal(d.get("density"), diag["density"])
    assert acc.max() > 0 and np.array_equal(d.get("accumulated_precipitation"), acc)
    assert len(set(dts)) == len(dts), "forced winds: every sub-step must have had its own dt"
    parity_record("trajectory", f"whole_step_loop/96x64x20/exact_mode_{n_cpu}_substeps", {n: {"bitdiff_cells": 0, "cells": int(s[n].size)} for n in s})
    d.close()


@pytest.mark.parametrize("exact", [False, True])
def test_config1_256x256x40_mpdata_thompson_substep(th_oracle, exact):
    """(exact: with icar_hip_mpdata_exact(ctx, 1) the advected fields are bit-identical too)
    BASELINE configs[1] at its literal size: one [Thompson -> MPDATA order 2 + FCT of the 9 scalars] step on 256 x 256 x 40.
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
    if exact: check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
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
        if exact: assert bits_equal(d.get(MEMBER[n]), s[n]), f"{n}: {nbitdiff(d.get(MEMBER[n]), s[n])} cells differ"
        else: assert_fields_close(d.get(MEMBER[n]), s[n], n, record=("trajectory", "config1/256x256x40/mpdata_after_thompson"))
    if exact: parity_record("trajectory", "config1/256x256x40/exact_mode_substep", {n: {"bitdiff_cells": 0, "cells": int(s[n].size)} for n in ADV_ORDER})
    d.close()


@pytest.mark.parametrize("exact", [False, True])
def test_config3_tile_update_winds_then_substep(th_oracle, oracle, exact):
    """The per-GPU workload of BASELINE configs[3] (1024 x 1024 x 40 on 2 x 4 images: a 512 x 256 x 40 tile): update_winds with
    windtype kWIND_LINEAR -- spatial_winds interpolating a look-up table (uploaded, 2 x 4 x 3 entries: the build is the init-time
    row W3, tests/test_gpu_winds.py) + balance_uvw -- then [Thompson -> MPDATA] with the new winds.  Winds and the microphysics
    bit for bit against the oracle chain (the C library's float functions on both sides), MPDATA on every cell to 1e-5."""
    from oracle import wind_oracle as W
    from icar_amd import linear_winds as LW
    from icar_amd.options import lt_options_type
    from icar_amd.wind import update_winds, kWIND_LINEAR
    from icar_amd.domain import domain_t
    from icar_amd.grid import grid_t
    from util import bits_equal
    nx, ny, nz = 512, 256, 40
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.5)).astype(np.float32)
    dxf = float(c["dx"])
    rng = np.random.default_rng(77)
    ndir, nspd, nnsq = 4, 3, 2
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.physics.windtype = kWIND_LINEAR
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = dxf
    opt.lt_options = lt_options_type(buffer=4, n_dir_values=ndir, n_spd_values=nspd, n_nsq_values=nnsq, stability_window_size=3,
                                     vert_smooth=2, variable_N=True, smooth_nsq=True, linear_contribution=0.5, linear_update_fraction=1.0)
    lt = opt.lt_options
    mp_var_request(opt)
    d = domain_t(grid_t().set_grid_dimensions(nx, ny, nz, 1, 1), device=0, dx=dxf)
    if exact: check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")     # MPDATA in the reference's operation order: bit for bit
    d.load_case(c)
    zc = (np.cumsum(c["dz_levels"]) - c["dz_levels"] / 2).astype(np.float32)
    z3 = np.ascontiguousarray(c["terrain"][:, None, :] + zc[None, :, None] * np.ones((ny, 1, nx), np.float32), np.float32)
    d.set("z", z3)
    mp_init(opt, d); adv_init(d, opt)
    LW.setup_linwinds(d, opt, c["terrain"], build=False)
    ulut = (0.5 * rng.standard_normal((ny, nz, nx + 1, nnsq, ndir, nspd))).astype(np.float32)
    vlut = (0.5 * rng.standard_normal((ny + 1, nz, nx, nnsq, ndir, nspd))).astype(np.float32)
    LW.lut_upload(d, opt, 0, ulut); LW.lut_upload(d, opt, 1, vlut)
    lo, hi = lt.resolved()
    dirv = W.linear_space(lt.dirmin, lt.dirmax, ndir); spdv = W.linear_space(lt.spdmin, lt.spdmax, nspd); nsqv = W.linear_space(lo, hi, nnsq)
    o = dict(variable_N=True, smooth_nsq=True, N_squared=lt.N_squared, max_stability=lt.max_stability, min_stability=lt.min_stability,
             linear_contribution=lt.linear_contribution, linear_update_fraction=lt.linear_update_fraction)
    hyd = tuple(c[k] for k in ("cloud_water", "cloud_ice", "rain", "snow"))
    oracle.set_math_mode(0)
    u, v = c["u"].copy(), c["v"].copy()
    up = np.zeros_like(u); vp = np.zeros_like(v)
    oracle.make_winds_grid_relative(u, v, np.zeros((ny, nx)), np.ones((ny, nx)))
    oracle.spatial_winds(u, v, c["potential_temperature"], c["exner"], z3, c["water_vapor"], hyd, ulut, vlut, up, vp, o, dirv, spdv, nsqv,
                         lt.vert_smooth, lt.stability_window_size)
    w = oracle.balance_uvw(u, v, c["jacobian_u"], c["jacobian_v"], c["jacobian_w"], c["advection_dz"], dxf)
    del ulut, vlut
    d.set("sintheta", np.zeros((ny, nx))); d.set("costheta", np.ones((ny, nx)))
    update_winds(d, opt)
    for n, want in (("u", u), ("v", v), ("w", w)):
        g = d.get(n)
        assert bits_equal(g, want), f"update_winds {n}: {(g != want).sum()} of {g.size} differ, max {abs(g - want).max()}"
    assert abs(u - c["u"]).max() > 0.05
    cw = dict(c); cw["u"], cw["v"], cw["w"] = u, v, w
    dt = float(np.float32(min(ideal.cfl_dt(cw), 60.0)))
    s = {n: c[n].copy() for n in ADV_ORDER}
    mp(d, opt, dt); d.model_time_seconds += dt
    _thompson_oracle_step(th_oracle, s, c, dt, nx, ny, nz)
    for n in ADV_ORDER:
        got = d.get(MEMBER[n])
        assert np.array_equal(got, s[n]), f"Thompson {n}: {(got != s[n]).sum()} cells differ"
    advect(d, opt, dt)
    _advect_oracle(th_oracle, s, cw, dt, ADV_ORDER)
    for n in ADV_ORDER:
        if exact: assert bits_equal(d.get(MEMBER[n]), s[n]), f"{n}: {nbitdiff(d.get(MEMBER[n]), s[n])} cells differ"
        else: assert_fields_close(d.get(MEMBER[n]), s[n], n, record=("trajectory", "config3_tile/512x256x40/mpdata_with_linear_winds"))
    d.close()
