"""Oracle vs the compiled reference (oracle/_ref/libicar_ref.so) on fresh seeded inputs: bit-exact.
Skipped where the reference build is absent.  One grid size per process for the reference's
advection (module-level SAVE arrays), so the advection checks share one size."""
import numpy as np
import pytest
from icar_amd import ideal
from util import bits_equal, nbitdiff, adv_args

ref = pytest.importorskip("oracle.ref")
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
NX, NY, NZ = 36, 28, 9


@pytest.mark.parametrize("scheme,dens,fct,seed", [(1, 0, 1, 1), (1, 1, 1, 2), (2, 0, 1, 3), (2, 1, 1, 4), (2, 0, 0, 5)])
def test_advect_matches_reference(oracle, scheme, dens, fct, seed):
    c = ideal.make_case(NX, NY, NZ, hill_height=900.0, noise=0.05, seed=seed, n_hydro=1, u0=8.0 + seed, v0=-4.0 + seed)
    dt = ideal.cfl_dt(c)
    names = ["water_vapor", "rain", "ice_number"]
    qa = np.stack([c[n] for n in names]).copy(); qb = qa.copy()
    ref.advect(scheme, qa, *adv_args(c), dt, advect_density=dens, fct=fct, nsteps=2)
    oracle.advect(scheme, qb, *adv_args(c), dt, advect_density=dens, fct=fct, nsteps=2)
    assert bits_equal(qa, qb), f"{nbitdiff(qa, qb)} values differ"


@pytest.mark.parametrize("order,dens,seed", [(2, 0, 21), (2, 1, 22), (3, 0, 23)])
def test_mpdata_rough_winds_match_reference(oracle, order, dens, seed):
    """white noise of 0.5 m/s on u and v, w rebalanced: the limiter is active on nearly every face and its all-or-nothing factor next
    to the ring (fin = fout = 0 there) decides cells on single ulps of the donor-cell pass -- the regime of round 6's bug hunt"""
    from util import roughen_winds
    c = ideal.make_case(NX, NY, NZ, hill_height=900.0, noise=0.05, seed=seed, n_hydro=1)
    c = roughen_winds(c, oracle, 0.5, seed=seed)
    dt = ideal.cfl_dt(c)
    names = ["water_vapor", "potential_temperature", "rain"]
    qa = np.stack([c[n] for n in names]).copy(); qb = qa.copy()
    ref.advect(2, qa, *adv_args(c), dt, advect_density=dens, mpdata_order=order, fct=1, nsteps=3)
    oracle.advect(2, qb, *adv_args(c), dt, advect_density=dens, mpdata_order=order, fct=1, nsteps=3)
    assert bits_equal(qa, qb), f"{nbitdiff(qa, qb)} values differ"


@pytest.mark.parametrize("order,dens,fct,seed", [(3, 0, 1, 6), (3, 1, 1, 7), (4, 0, 1, 8), (3, 0, 0, 9), (1, 0, 1, 10)])
def test_mpdata_order_matches_reference(oracle, order, dens, fct, seed):
    """adv_mpdata.f90:372-402, the iord loop beyond the default order 2 (q2 = q before every further corrective
    iteration, limiter against the field that iteration started from), and mpdata_order 1 (= donor cell): bit-exact."""
    c = ideal.make_case(NX, NY, NZ, hill_height=900.0, noise=0.05, seed=seed, n_hydro=1, u0=8.0 + seed % 5, v0=-4.0 + seed % 5)
    dt = ideal.cfl_dt(c)
    names = ["water_vapor", "rain", "ice_number"]
    qa = np.stack([c[n] for n in names]).copy(); qb = qa.copy()
    ref.advect(2, qa, *adv_args(c), dt, advect_density=dens, mpdata_order=order, fct=fct, nsteps=2)
    oracle.advect(2, qb, *adv_args(c), dt, advect_density=dens, mpdata_order=order, fct=fct, nsteps=2)
    assert bits_equal(qa, qb), f"{nbitdiff(qa, qb)} values differ"
    if order >= 3:                                      # the extra iteration must have done something
        qc = np.stack([c[n] for n in names]).copy()
        oracle.advect(2, qc, *adv_args(c), dt, advect_density=dens, mpdata_order=2, fct=fct, nsteps=2)
        assert not bits_equal(qb, qc)


@pytest.mark.parametrize("seed,moist,cool", [(11, 1.6, 0.4), (12, 2.5, 1.5), (13, 0.7, 0.0)])
def test_mp_simple_matches_reference(oracle, seed, moist, cool):
    nx, ny, nz = 33, 21, 25
    c = ideal.make_case(nx, ny, nz, hill_height=700.0, noise=0.03, seed=seed)
    keys = ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain", "snow", "dz_mass"]
    a = {k: c[k].copy() for k in keys}; a["water_vapor"] = (a["water_vapor"] * np.float32(moist)).astype(np.float32)
    b = {k: v.copy() for k, v in a.items()}
    ra = np.zeros((ny, nx), np.float32); sa = ra.copy(); rb = ra.copy(); sb = ra.copy()
    oracle.set_math_mode(0)
    for _ in range(6):
        ref.mp_simple(*[a[k] for k in keys[:8]], ra, sa, 45.0, a["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
        assert oracle.mp_simple(*[b[k] for k in keys[:8]], rb, sb, 45.0, b["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz) == 0
        a["potential_temperature"] -= np.float32(cool); b["potential_temperature"] -= np.float32(cool)
    for k in keys:
        assert bits_equal(a[k], b[k]), f"{k}: {nbitdiff(a[k], b[k])} values differ"
    assert bits_equal(ra, rb) and bits_equal(sa, sb)
