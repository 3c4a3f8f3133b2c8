"""The scenario of the reference's CI run (tests/gen_ideal_test.py with Schaer_test=True): the
terrain-following advection test of Schaer et al. 2002 -- 300 x 20 x 50 cells, dx = 1 km, dz = 500 m, a 3 km wavy mountain,
10 m/s aloft and calm air below 4 km, a tracer anomaly upstream -- with adv = 1 (upwind), mp = 0 and wind = 3
(iterative_winds).  The reference's CI only asserts that a file with more than one time step comes out; here the device
path runs update_winds(windtype 3) + the step loop and is compared bit-for-bit with the CPU oracle doing the same, plus the
properties the scenario is about: the anomaly crosses the mountain, stays bounded by its initial extrema (donor-cell
monotonicity) and the domain keeps its tracer mass until the anomaly reaches the outflow boundary."""
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.options import options_t
from util import single_image_domain, bits_equal

pytestmark = pytest.mark.gpu
NX, NY, NZ, DX, DZ = 300, 20, 50, 1000.0, 500.0


def schaer_case():
    x = (np.arange(NX) - NX / 2 + 0.5) * DX
    h = 3000.0 * np.exp(-(x / 25000.0) ** 2) * np.cos(np.pi * x / 8000.0) ** 2                       # Schaer 2002 eq. 25
    c = ideal.make_case(NX, NY, NZ, dx=DX, uniform_dz=DZ, terrain=np.broadcast_to(h[None, :], (NY, NX)), u0=0.0, v0=0.0)
    zc = (np.arange(NZ) + 0.5) * DZ
    z = c["terrain"][:, None, :] + zc[None, :, None] * c["jacobian"]                                  # mass-level heights
    zu = np.concatenate([z[:, :, :1], 0.5 * (z[:, :, 1:] + z[:, :, :-1]), z[:, :, -1:]], axis=2)
    prof = np.where(zu >= 5000.0, 1.0, np.where(zu <= 4000.0, 0.0, np.sin(0.5 * np.pi * (zu - 4000.0) / 1000.0) ** 2))
    c["u"] = (10.0 * prof).astype(np.float32)                                                         # eq. 26
    c["v"] = np.zeros_like(c["v"])
    xx = x[None, None, :]
    r = np.sqrt(((xx + 50000.0) / 25000.0) ** 2 + ((z - 9000.0) / 3000.0) ** 2)
    c["water_vapor"] = np.where(r <= 1.0, np.cos(0.5 * np.pi * r) ** 2, 0.0).astype(np.float32)      # eq. 27
    return c, z


def test_schaer_advection_scenario(oracle):
    from icar_amd.wind import update_winds, kITERATIVE_WINDS
    from icar_amd.time_step import step, update_dt
    from icar_amd.advection import adv_init
    from icar_amd.constants import kADV_UPWIND
    c, z = schaer_case()
    opt = options_t()
    opt.physics.advection = kADV_UPWIND; opt.physics.microphysics = 0; opt.physics.windtype = kITERATIVE_WINDS
    opt.parameters.wind_iterations = 100
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = DX
    opt.vars_to_advect = {"water_vapor": 1}
    d = single_image_domain(c)
    d.exchange_vars = ["water_vapor"]
    adv_init(d, opt)
    update_winds(d, opt)                                                      # make_winds_grid_relative + iterative_winds + balance_uvw on the device
    geo = (c["jacobian_u"], c["jacobian_v"], c["jacobian_w"], c["advection_dz"], c["jacobian"], DX)
    u0, v0 = c["u"].copy(), c["v"].copy()
    ny_, nx_ = c["jacobian"].shape[0], c["jacobian"].shape[2]
    oracle.make_winds_grid_relative(u0, v0, np.zeros((ny_, nx_)), np.ones((ny_, nx_)))     # unrotated grid (wind.f90:300)
    u, v, _ = oracle.iterative_winds(u0, v0, *geo, 100)
    w = oracle.balance_uvw(u, v, *geo[:4], DX)
    assert bits_equal(d.get("u"), u) and bits_equal(d.get("v"), v) and bits_equal(d.get("w"), w)
    div = oracle.calc_divergence(u, v, w, *geo)
    assert np.abs(div[1:-1, :, 2:-2]).max() < 1e-6                            # balance_uvw closes the column budget
    dt = update_dt(d, opt)
    nsteps = 60
    end_time = (nsteps - 0.5) * dt                                            # the last sub-step is clamped (time_step.f90:469-471)
    n = step(d, end_time, opt, diagnostics=False)
    assert n == nsteps
    q = c["water_vapor"][None].copy()
    t = 0.0
    while t < end_time:
        dti = dt if t + dt <= end_time else end_time - t
        oracle.advect(1, q, u, v, w, c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                      c["advection_dz"], c["dz_levels"], DX, float(np.float32(dti)))
        t += dti
    got = d.get("water_vapor")
    assert bits_equal(got, q[0]), f"{(got != q[0]).sum()} cells differ after {nsteps} steps"
    # the scenario's own properties
    q0 = c["water_vapor"]
    assert got.min() >= 0.0 and got.max() <= q0.max()                         # donor cell: no new extrema
    vol = c["jacobian"] * c["advection_dz"]
    m0, m1 = float((q0 * vol).sum(dtype=np.float64)), float((got * vol).sum(dtype=np.float64))
    assert abs(m1 - m0) < 2e-3 * m0                                           # nothing has left the domain yet
    xc = (np.arange(NX) + 0.5)[None, None, :]
    c0 = float((q0 * vol * xc).sum(dtype=np.float64) / m0); c1 = float((got * vol * xc).sum(dtype=np.float64) / m1)
    moved_km = (c1 - c0)
    assert 0.8 * 10.0 * end_time / 1000.0 < moved_km < 1.2 * 10.0 * end_time / 1000.0, (moved_km, end_time)
    d.close()
