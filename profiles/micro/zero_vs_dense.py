"""MPDATA time when the 7 hydrometeor scalars are all zero vs dense (non-zero everywhere), skipping off / on."""
import sys, os, time, ctypes
sys.path.insert(0, ".")
import numpy as np
from icar_amd import ideal, capi
from icar_amd.options import options_t
from icar_amd.grid import grid_t
from icar_amd.domain import domain_t
from icar_amd.microphysics import mp_var_request
from icar_amd.advection import advect
from icar_amd.constants import kMP_THOMPSON, kADV_MPDATA
nx = ny = 512; nz = 40
c = ideal.make_case(nx, ny, nz, hill_height=1000., noise=0.01, n_hydro=1)
mode = sys.argv[1]
rng = np.random.default_rng(0)
for k in ("cloud_water", "rain", "snow", "cloud_ice", "graupel", "ice_number", "rain_number"):
    c[k] = np.zeros_like(c[k]) if mode == "zero" else (1e-4 * (1 + rng.random(c[k].shape))).astype(np.float32)
opt = options_t(); opt.physics.microphysics = kMP_THOMPSON; opt.physics.advection = kADV_MPDATA; mp_var_request(opt)
d = domain_t(grid_t().set_grid_dimensions(nx, ny, nz, 1, 1)); d.load_case(c)
dt = ideal.cfl_dt(c)
for it in range(3): advect(d, opt, dt)
d.synchronize(); t = time.time()
for it in range(5): advect(d, opt, dt)
d.synchronize(); print(mode, "skip" if not os.environ.get("ICAR_HIP_MPDATA_NO_SKIP") else "noskip", "advect ms/call", round((time.time() - t) / 5 * 1e3, 3))
