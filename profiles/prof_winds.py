"""Timing of rows W3 (LUT build) and W2 (spatial_winds) at the north-star tile: 512x512 global terrain, 40 levels,
the reference's default LUT axes (24 dir x 6 spd x 5 N^2 = 720 combos), buffer 50 -> 616x616 FFTs.
usage: python profiles/prof_winds.py [n] [nz]"""
import sys, time, ctypes
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import linear_winds as LW
from icar_amd.capi import lib, check
from icar_amd.domain import domain_t
from icar_amd.grid import grid_t
from icar_amd.options import options_t
from wind_case import terrain, atmosphere

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 40
opt = options_t()
dz = np.array([50., 75., 125., 200., 300., 400.] + [500.] * 34, np.float32)[:nz]
opt.parameters.dz_levels = dz
g = grid_t().set_grid_dimensions(n, n, nz, 1, 1)
d = domain_t(g, device=0, dx=2000.0)
t = terrain(n, n)
t0 = time.time(); LW.setup_linwinds(d, opt, t, build=False); d.synchronize(); t_setup = time.time() - t0
zc = np.cumsum(dz, dtype=np.float32) - dz / np.float32(2)
zb, zt = LW.layer_bounds(zc, 0.0, dz)
nsub = sum(max(1, int(np.ceil((b - a) / 100.0))) for a, b in zip(zb, zt))
t0 = time.time(); LW.build_lut(d, zb, zt); d.synchronize(); t_lut = time.time() - t0
a = atmosphere(n, n, nz, seed=2)
for k in ("z", "potential_temperature", "exner", "water_vapor", "cloud_water_mass", "cloud_ice_mass", "rain_mass", "snow_mass", "u", "v"):
    d.set(k, a[k])
LW.linear_perturb(d, opt); d.synchronize()
check(lib().icar_hip_timing_enable(d.ctx, 1), "timing")
reps = 5
t0 = time.time()
for _ in range(reps):
    LW.linear_perturb(d, opt)
d.synchronize(); t_sw = (time.time() - t0) / reps
lt = opt.lt_options
ncombo = lt.n_dir_values * lt.n_spd_values * lt.n_nsq_values
lut_gb = ncombo * 4.0 * ((n + 1) * nz * n + n * nz * (n + 1)) / 1e9
print({"global": [n, n, nz], "fft": [n + 104, n + 104], "combos": ncombo, "lut_GB": round(lut_gb, 1),
       "setup_s": round(t_setup, 3), "lut_build_s": round(t_lut, 3), "reference_inverse_ffts": 2 * ncombo * nsub,
       "device_inverse_ffts": 2 * (ncombo - ncombo // lt.n_spd_values) * nz, "spatial_winds_ms": round(1e3 * t_sw, 3),
       "u_abs_max": float(abs(d.get("u")).max())})
