"""CPU checks of the linear-wind oracle (oracle/wind_oracle.py + wind_oracle.c; rows W2/W3, PARITY UNPINNED:
no executable reference and no expected values in the reference's tests).  What can be pinned is pinned here:
the tests/test_fftshift.f90 scenario, and closed-form properties of the restated routines."""
import numpy as np
import pytest
from oracle import wind_oracle as W
from icar_amd.options import lt_options_type
from wind_case import terrain, lut_options, atmosphere


def test_fftshift_scenario_of_reference_test():
    # tests/test_fftshift.f90: n=5, x(i,j) = i + 50 j; reverted == original; shift is by (n+1)/2 (F9)
    x = np.array([[i + 50 * j for j in range(1, 6)] for i in range(1, 6)], np.float32)
    s = W.fftshift2r(x)
    assert np.array_equal(s[:, 0], [153, 154, 155, 151, 152])
    assert np.array_equal(W.ifftshift2r(s), x)
    xc = x.astype(np.complex128) * (1 + 0.5j)
    assert np.array_equal(W.ifftshift2cc(W.fftshift2cc(xc)), xc)         # small integers survive the single-precision temp
    big = np.full((4, 4), 1.0 + 2.0 ** -30, np.complex128)               # F9: values are rounded to single precision
    assert np.all(W.fftshift2cc(big) == 1.0)
    for n in (4, 5, 8, 9):                                               # even n: the usual half shift
        v = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 1), np.float32)
        assert np.array_equal(W.ifftshift2r(W.fftshift2r(v)), v)


def test_add_buffer_topo_properties():
    t = terrain(30, 22).T.copy()        # [nx, ny]
    b = 6
    bt = W.add_buffer_topo(t, 3, b)
    assert bt.shape == (30 + 2 * b, 22 + 2 * b) and np.all(bt.imag == 0)
    assert np.array_equal(bt[b:-b, b:-b].real.astype(np.float32), t)     # interior untouched
    # flat terrain stays flat (blend weights sum to 1, box means of a constant)
    flat = W.add_buffer_topo(np.full((12, 9), 321.5, np.float32), 5, 4)
    np.testing.assert_allclose(flat.real, 321.5, rtol=1e-7)
    # outermost blended ring is periodic-compatible: first/last x rows tend to the mean of both edges
    nb = W.add_buffer_topo(t, 0, b)
    np.testing.assert_allclose(nb[0, b:-b].real, 0.5 * (t[0, :] + t[-1, :]), rtol=1e-6)
    np.testing.assert_allclose(nb[-1, b:-b].real, 0.5 * (t[0, :] + t[-1, :]), rtol=1e-6)


def test_linear_perturbation_properties():
    t = terrain(28, 24).T.copy()
    tf, lt, buf = W.setup_linwinds(t, 2000.0, 8)
    assert buf == 10 and tf.shape == (28 + 20, 24 + 20)
    assert np.array_equal(tf, tf.astype(np.complex64).astype(np.complex128))      # F9 rounding happened
    z0, z1 = 150.0, 420.0
    u0, v0 = W.linear_perturbation_constz(0.0, 0.0, 1e-4, z0, z1, 100.0, tf, lt)
    assert not u0.any() and not v0.any()                                  # :248-252
    assert W.n_steps_of(z0, z1, 100.0) == 3 and W.n_steps_of(0, 50, 100) == 1
    up, vp = W.linear_perturbation_constz(9.0, 4.0, 1e-4, z0, z1, 100.0, tf, lt)
    assert np.isfinite(up).all() and np.isfinite(vp).all() and abs(up.real).max() > 0.1
    # linear in the terrain spectrum (up to the single-precision ifftshift temp)
    up2, vp2 = W.linear_perturbation_constz(9.0, 4.0, 1e-4, z0, z1, 100.0, 2 * tf, lt)
    np.testing.assert_allclose(up2.real, 2 * up.real, atol=2e-6 * abs(up.real).max())
    # average of the sub-layer solutions
    acc = 0
    for zc in (195.0, 285.0, 375.0):
        a, _ = W.linear_perturbation_at_height(9.0, 4.0, 1e-4, zc, tf, lt)
        acc = acc + a
    np.testing.assert_allclose((acc / 3.0).real, up.real, atol=1e-9 * abs(up.real).max())
    # constant-z is the varying-z form with flat bounds (same sub-layer centres when dz is a multiple of the step)
    nxg, nyg = t.shape
    uv, vv = W.linear_perturbation_varyingz(9.0, 4.0, 1e-4, np.full((nxg, nyg), 100.0, np.float32), np.full((nxg, nyg), 400.0, np.float32),
                                            100.0, tf, lt, buf)
    uc, vc = W.linear_perturbation_constz(9.0, 4.0, 1e-4, 100.0, 400.0, 100.0, tf, lt)
    np.testing.assert_allclose(uv.real, uc.real, atol=1e-9 * abs(uc.real).max())


def test_build_lut_small():
    lt_o = lt_options_type(buffer=5, n_dir_values=4, n_spd_values=3, n_nsq_values=2)
    t = terrain(14, 12).T.copy()
    tf, lt, buf = W.setup_linwinds(t, 1500.0, lt_o.buffer)
    zb = np.array([0.0, 120.0], np.float32); zt = np.array([120.0, 390.0], np.float32)
    ul, vl, dirv, spdv, nsqv = W.build_lut(tf, lt, buf, zb, zt, lut_options(lt_o))
    assert ul.shape == (3, 4, 2, 15, 2, 12) and vl.shape == (3, 4, 2, 14, 2, 13)
    assert spdv[0] == 0 and not ul[0].any() and not vl[0].any()          # zero speed -> zero perturbation
    assert np.isfinite(ul).all() and abs(ul[2]).max() > 0.01
    assert dirv[0] == 0 and abs(dirv[-1] - 2 * np.pi) < 1e-6 and np.all(np.diff(nsqv) > 0)


def test_spatial_winds_oracle_properties(oracle):
    nx, ny, nz = 26, 17, 9
    a = atmosphere(nx, ny, nz)
    lt_o = lt_options_type(n_dir_values=6, n_spd_values=4, n_nsq_values=3, stability_window_size=3, vert_smooth=2)
    lo, hi = lt_o.resolved()
    dirv = W.linear_space(lt_o.dirmin, lt_o.dirmax, 6); spdv = W.linear_space(0, 30, 4); nsqv = W.linear_space(lo, hi, 3)
    cval = np.float32(0.75)
    ulut = np.full((ny, nz, nx + 1, 3, 6, 4), cval, np.float32); vlut = np.full((ny + 1, nz, nx, 3, 6, 4), -cval, np.float32)
    opt = dict(variable_N=True, smooth_nsq=True, N_squared=3e-5, max_stability=6e-4, min_stability=1e-7,
               linear_contribution=1.0, linear_update_fraction=0.2)
    u = a["u"].copy(); v = a["v"].copy()
    up = np.zeros_like(u); vp = np.zeros_like(v)
    hyd = (a["cloud_water_mass"], a["cloud_ice_mass"], a["rain_mass"], a["snow_mass"])
    nsq = oracle.spatial_winds(u, v, a["potential_temperature"], a["exner"], a["z"], a["water_vapor"], hyd, ulut, vlut, up, vp,
                               opt, dirv, spdv, nsqv, lt_o.vert_smooth, lt_o.stability_window_size)
    # a constant LUT interpolates to the constant whatever the brackets; relaxation with f=0.2 from zero
    np.testing.assert_allclose(up, 0.2 * cval, rtol=3e-6)
    np.testing.assert_allclose(vp, -0.2 * cval, rtol=3e-6)
    np.testing.assert_allclose(u - a["u"], up, atol=2e-6)
    assert nsq.min() >= 1e-7 * (1 - 1e-5) and nsq.max() <= 6e-4 * (1 + 1e-5)
    # smooth_array(ydim=3) keeps a constant field constant and preserves the mean roughly
    c = np.full((ny, nz, nx), 2.5, np.float32)
    assert np.allclose(oracle.smooth_array_ydim3(c.copy(), 3), 2.5, rtol=1e-7)
    # calc_direction quadrants (atm_utilities.f90:334-355)
    assert abs(oracle.calc_direction(1.0, 1.0) - np.pi / 4) < 1e-6
    assert abs(oracle.calc_direction(1.0, 0.0) - np.pi / 2) < 1e-6
    assert abs(oracle.calc_direction(0.0, -1.0) - np.pi) < 1e-6
    assert abs(oracle.calc_direction(-1.0, 0.0) - 1.5 * np.pi) < 1e-6
    assert abs(oracle.calc_direction(-1.0, 1.0) - 1.75 * np.pi) < 1e-6
