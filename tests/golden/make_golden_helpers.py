#!/usr/bin/env python
"""Golden vectors of the helper procedures rows T3 / W2 / W3 call (utilities/atm_utilities.f90, utilities/array_utilities.f90
compiled unmodified into oracle/_ref): exner_function, calc_direction / calc_speed / calc_u / calc_v, calc_stability (dry and
moist branch), compute_ivt / compute_iq, linear_space, calc_weight, smooth_array_3d.  The seeded inputs are rebuilt by inputs()
(shared with tests/test_oracle_helpers_golden.py); the reference's outputs are stored in tests/golden/helpers.npz.
Only runs where /root/reference is present."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def inputs():
    I = {}
    rng = np.random.default_rng(100)
    I["exner_p"] = rng.uniform(5000.0, 108000.0, (6, 11, 40)).astype(np.float32)
    rng = np.random.default_rng(101)
    u = rng.normal(0, 12, 4000).astype(np.float32); v = rng.normal(0, 12, 4000).astype(np.float32)
    u[:8] = [0, 0, 3, -3, 0, 5, -5, 1e-30]; v[:8] = [0, 4, 0, 0, -4, 1e-30, -1e-30, 0]
    I["polar_u"], I["polar_v"] = u, v
    rng = np.random.default_rng(102)
    n = 5000
    th_b = rng.uniform(270, 320, n).astype(np.float32); th_t = (th_b + rng.normal(1.0, 2.0, n)).astype(np.float32)
    pii_b = rng.uniform(0.7, 1.0, n).astype(np.float32); pii_t = (pii_b - rng.uniform(0.001, 0.02, n)).astype(np.float32)
    z_b = rng.uniform(0, 8000, n).astype(np.float32); z_t = (z_b + rng.uniform(20, 600, n)).astype(np.float32)
    qv_b = rng.uniform(1e-4, 0.02, n).astype(np.float32); qv_t = (qv_b * rng.uniform(0.8, 1.0, n)).astype(np.float32)
    qc = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(1e-8, 1e-3, n)).astype(np.float32)
    I["stab"] = (th_t, th_b, pii_t, pii_b, z_t, z_b, qv_t, qv_b, qc)
    rng = np.random.default_rng(103)
    ny, nz, nx = 7, 24, 19
    p_i = np.sort(rng.uniform(20000.0, 101000.0, (ny, nz, nx)).astype(np.float32), axis=1)[:, ::-1, :].copy()
    p_i[0, :, 0] = np.linspace(49000, 30000, nz)
    I["col"] = (rng.uniform(0, 0.02, (ny, nz, nx)).astype(np.float32), rng.normal(0, 10, (ny, nz, nx)).astype(np.float32),
                rng.normal(0, 10, (ny, nz, nx)).astype(np.float32), p_i)
    I["axes"] = ((0.0, 2 * np.pi, 24), (0.0, 30.0, 6), (np.log(1e-7), np.log(6e-4), 5), (-3.0, 7.5, 2))
    rng = np.random.default_rng(104)
    match = rng.uniform(-5, 40, 300).astype(np.float32); match[:3] = [0.0, 30.0, 6.0]
    I["match"] = match
    rng = np.random.default_rng(105)
    I["smooth"] = [(rng.normal(0, 1, shape).astype(np.float32), w) for shape, w in (((14, 5, 17), 2), ((30, 3, 9), 4), ((6, 4, 25), 7))]
    return I


if __name__ == "__main__":
    from oracle import ref
    I = inputs(); out = {}
    out["exner"] = ref.exner(I["exner_p"]).reshape(I["exner_p"].shape)
    d, s, ub, vb = ref.wind_polar(I["polar_u"], I["polar_v"])
    out["polar_dir"], out["polar_speed"], out["polar_u_back"], out["polar_v_back"] = d, s, ub, vb
    out["stability"] = ref.calc_stability(*I["stab"])
    qv, u, v, p_i = I["col"]
    out["ivt"] = ref.compute_ivt(qv, u, v, p_i); out["iq"] = ref.compute_iq(qv, p_i)
    for m, (lo, hi, n) in enumerate(I["axes"]):
        out[f"axis{m}"] = ref.linear_space(lo, hi, n)
    axis = out["axis1"]
    best = np.array([max(1, int(np.sum(x > axis))) for x in I["match"]], np.int32)
    out["weight_next"], out["weight"] = ref.calc_weight(axis, best, I["match"])
    for m, (a, w) in enumerate(I["smooth"]):
        out[f"smooth{m}"] = ref.smooth_array_3d(a.copy(), w, 3)
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    print("wrote helpers.npz", {k: v.shape for k, v in out.items()})
