"""The helper procedures rows T3 and W2 call, oracle restatement vs the UNMODIFIED reference modules compiled into
oracle/_ref (utilities/atm_utilities.f90, utilities/array_utilities.f90): bit-exact in the oracle's libm mode.
This pins by execution: exner_function (T3), compute_ivt / compute_iq (T3 optional integrals), calc_direction /
calc_speed / calc_u / calc_v, calc_stability (dry and moist branch), linear_space, calc_weight and smooth_array_3d
(all W2 / W3 axes).  spatial_winds' own body (linear_winds.f90, needs FFTW) stays restated from the source."""
import numpy as np
import pytest
from util import bits_equal, nbitdiff

ref = pytest.importorskip("oracle.ref")
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def test_exner_function(oracle):
    oracle.set_math_mode(0)
    rng = np.random.default_rng(0)
    p = rng.uniform(5000.0, 108000.0, (6, 11, 40)).astype(np.float32)
    th = np.full_like(p, 300.0); z = np.zeros((6, 11, 41), np.float32); zv = np.zeros((7, 11, 40), np.float32)
    got = oracle.diagnostic_update(p, th, z, zv, np.zeros_like(p), z, zv, np.ones_like(p))["exner"]
    assert bits_equal(got, ref.exner(p).reshape(p.shape))


def test_wind_polar_helpers(oracle):
    from oracle import wind_oracle as W
    oracle.set_math_mode(0)
    rng = np.random.default_rng(1)
    u = rng.normal(0, 12, 4000).astype(np.float32); v = rng.normal(0, 12, 4000).astype(np.float32)
    u[:8] = [0, 0, 3, -3, 0, 5, -5, 1e-30]; v[:8] = [0, 4, 0, 0, -4, 1e-30, -1e-30, 0]
    d_ref, s_ref, ub, vb = ref.wind_polar(u, v)
    d = np.array([oracle.calc_direction(a, b) for a, b in zip(u, v)], np.float32)
    assert bits_equal(d, d_ref)
    assert bits_equal(np.sqrt(u * u + v * v), s_ref)
    assert bits_equal(np.array([W.calc_u(a, b) for a, b in zip(d_ref, s_ref)], np.float32), ub)
    assert bits_equal(np.array([W.calc_v(a, b) for a, b in zip(d_ref, s_ref)], np.float32), vb)


def test_calc_stability_dry_and_moist(oracle):
    oracle.set_math_mode(0)
    rng = np.random.default_rng(2)
    n = 5000
    th_b = rng.uniform(270, 320, n).astype(np.float32); th_t = (th_b + rng.normal(1.0, 2.0, n)).astype(np.float32)
    pii_b = rng.uniform(0.7, 1.0, n).astype(np.float32); pii_t = (pii_b - rng.uniform(0.001, 0.02, n)).astype(np.float32)
    z_b = rng.uniform(0, 8000, n).astype(np.float32); z_t = (z_b + rng.uniform(20, 600, n)).astype(np.float32)
    qv_b = rng.uniform(1e-4, 0.02, n).astype(np.float32); qv_t = (qv_b * rng.uniform(0.8, 1.0, n)).astype(np.float32)
    qc = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(1e-8, 1e-3, n)).astype(np.float32)
    a = (th_t, th_b, pii_t, pii_b, z_t, z_b, qv_t, qv_b, qc)
    want = ref.calc_stability(*a)
    got = oracle.calc_stability(*a)
    assert bits_equal(got, want), nbitdiff(got, want)
    assert (qc < 1e-7).any() and (qc >= 1e-7).any()


def test_column_integrals(oracle):
    rng = np.random.default_rng(3)
    ny, nz, nx = 7, 24, 19
    p_i = np.sort(rng.uniform(20000.0, 101000.0, (ny, nz, nx)).astype(np.float32), axis=1)[:, ::-1, :].copy()
    p_i[0, :, 0] = np.linspace(49000, 30000, nz)          # a column entirely above 500 hPa
    qv = rng.uniform(0, 0.02, (ny, nz, nx)).astype(np.float32)
    u = rng.normal(0, 10, (ny, nz, nx)).astype(np.float32); v = rng.normal(0, 10, (ny, nz, nx)).astype(np.float32)
    assert bits_equal(oracle.compute_ivt(qv, u, v, p_i), ref.compute_ivt(qv, u, v, p_i))
    assert bits_equal(oracle.compute_iq(qv, p_i), ref.compute_iq(qv, p_i))
    assert ref.compute_iq(qv, p_i)[0, 0] == 0 and ref.compute_iq(qv, p_i).max() > 1


def test_lut_axes_and_weights(oracle):
    from oracle import wind_oracle as W
    for lo, hi, n in ((0.0, 2 * np.pi, 24), (0.0, 30.0, 6), (np.log(1e-7), np.log(6e-4), 5), (-3.0, 7.5, 2)):
        assert bits_equal(W.linear_space(lo, hi, n), ref.linear_space(lo, hi, n)), (lo, hi, n)
    axis = ref.linear_space(0.0, 30.0, 6)
    rng = np.random.default_rng(4)
    match = rng.uniform(-5, 40, 300).astype(np.float32); match[:3] = [0.0, 30.0, 6.0]
    best = np.array([max(1, int(np.sum(m > axis))) for m in match], np.int32)        # the bracket search of spatial_winds
    n_ref, w_ref = ref.calc_weight(axis, best, match)
    n_o, w_o = oracle.calc_weight(axis, best, match)
    assert np.array_equal(n_ref, n_o) and bits_equal(w_ref, w_o)


@pytest.mark.parametrize("shape,w", [((14, 5, 17), 2), ((30, 3, 9), 4), ((6, 4, 25), 7)])
def test_smooth_array_3d(oracle, shape, w):
    rng = np.random.default_rng(5)
    a = rng.normal(0, 1, shape).astype(np.float32)
    want = ref.smooth_array_3d(a.copy(), w, 3)
    got = oracle.smooth_array_ydim3(a.copy(), w)
    assert bits_equal(got, want), nbitdiff(got, want)
