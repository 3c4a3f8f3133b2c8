"""The oracle's restatements of the helper procedures rows T3 / W2 / W3 call against golden vectors made by the reference's own
compiled modules (tests/golden/make_golden_helpers.py -> helpers.npz), bit for bit.  tests/test_oracle_helpers_vs_ref.py makes
the same comparisons by calling oracle/_ref directly where /root/reference is present; these fixtures hold everywhere."""
import os
import sys

import numpy as np
from util import bits_equal, nbitdiff

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_golden_helpers as G  # noqa: E402


def test_helpers_golden(oracle):
    from oracle import wind_oracle as W
    g = np.load(os.path.join(GOLD, "helpers.npz")); I = G.inputs()
    oracle.set_math_mode(0)
    p = I["exner_p"]; th = np.full_like(p, 300.0); z = np.zeros((6, 11, 41), np.float32); zv = np.zeros((7, 11, 40), np.float32)
    assert bits_equal(oracle.diagnostic_update(p, th, z, zv, np.zeros_like(p), z, zv, np.ones_like(p))["exner"], g["exner"])
    u, v = I["polar_u"], I["polar_v"]
    d = np.array([oracle.calc_direction(a, b) for a, b in zip(u, v)], np.float32)
    assert bits_equal(d, g["polar_dir"]) and bits_equal(np.sqrt(u * u + v * v), g["polar_speed"])
    assert bits_equal(np.array([W.calc_u(a, b) for a, b in zip(g["polar_dir"], g["polar_speed"])], np.float32), g["polar_u_back"])
    assert bits_equal(np.array([W.calc_v(a, b) for a, b in zip(g["polar_dir"], g["polar_speed"])], np.float32), g["polar_v_back"])
    got = oracle.calc_stability(*I["stab"])
    assert bits_equal(got, g["stability"]), nbitdiff(got, g["stability"])
    qv, uu, vv, p_i = I["col"]
    assert bits_equal(oracle.compute_ivt(qv, uu, vv, p_i), g["ivt"]) and bits_equal(oracle.compute_iq(qv, p_i), g["iq"])
    for m, (lo, hi, n) in enumerate(I["axes"]):
        assert bits_equal(W.linear_space(lo, hi, n), g[f"axis{m}"]), (lo, hi, n)
    axis = g["axis1"]
    best = np.array([max(1, int(np.sum(x > axis))) for x in I["match"]], np.int32)
    n_o, w_o = oracle.calc_weight(axis, best, I["match"])
    assert np.array_equal(n_o, g["weight_next"]) and bits_equal(w_o, g["weight"])
    for m, (a, w) in enumerate(I["smooth"]):
        got = oracle.smooth_array_ydim3(a.copy(), w)
        assert bits_equal(got, g[f"smooth{m}"]), nbitdiff(got, g[f"smooth{m}"])
