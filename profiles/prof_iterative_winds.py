"""Timing of iterative_winds at the north-star tile 512x512x40, wind_iterations = 100 (the
reference's default, options_obj.f90:1029), one image; the CPU oracle's sweep timed beside it on a bounded sample.
usage: python profiles/prof_iterative_winds.py [n] [nz] [iterations]"""
import sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.wind import iterative_winds
from util import single_image_domain
from oracle import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 40
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
c = ideal.make_case(n, n, nz, hill_height=1000.0, noise=0.02, n_hydro=1)
rng = np.random.default_rng(0)
c["u"] = (c["u"] + rng.standard_normal(c["u"].shape).astype(np.float32)).astype(np.float32)
c["v"] = (c["v"] + rng.standard_normal(c["v"].shape).astype(np.float32)).astype(np.float32)
d = single_image_domain(c)
opt = options_t(); opt.parameters.wind_iterations = iters
u0, v0 = c["u"].copy(), c["v"].copy()
iterative_winds(d, opt); d.synchronize()
reps = 3; t = 0.0
for _ in range(reps):
    d.set("u", u0); d.set("v", v0); d.synchronize()
    t0 = time.time(); iterative_winds(d, opt); d.synchronize(); t += time.time() - t0
t /= reps
geo = (c["jacobian_u"], c["jacobian_v"], c["jacobian_w"], c["advection_dz"], c["jacobian"], float(c["dx"]))
u, v = u0.copy(), v0.copy(); w = orc.balance_uvw(u, v, *geo[:4], geo[5]); orc.iterative_winds_correct_w(w, geo[3])
t0 = time.time()
for _ in range(3): orc.iterative_winds_sweep(u, v, w, *geo)
t_cpu = (time.time() - t0) / 3
cells = n * n * nz
sweep_ms = 1e3 * t / (iters + 1)
print(json.dumps({"tile": [n, n, nz], "wind_iterations": iters, "device_total_ms": round(1e3 * t, 3), "device_ms_per_sweep": round(sweep_ms, 4),
                  "algorithmic_bytes_per_cell_sweep": 60, "achieved_GBps": round(60 * cells / (sweep_ms * 1e-3) / 1e9, 1),
                  "cpu_oracle_ms_per_sweep": round(1e3 * t_cpu, 2), "cpu_cores": 1}))
