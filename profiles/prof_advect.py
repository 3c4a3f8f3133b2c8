"""Profile driver: N MPDATA advection steps of the 9 Thompson scalars on a 512x512x40 tile (no microphysics)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.grid import grid_t
from icar_amd.domain import domain_t
from icar_amd.microphysics import mp_var_request
from icar_amd.advection import advect
from icar_amd.constants import kMP_THOMPSON, kADV_MPDATA
nx = ny = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = ideal.make_case(nx, ny, nz, hill_height=1000., noise=0.01, n_hydro=1)
opt = options_t(); opt.physics.microphysics = kMP_THOMPSON; opt.physics.advection = kADV_MPDATA; mp_var_request(opt)
d = domain_t(grid_t().set_grid_dimensions(nx, ny, nz, 1, 1)); d.load_case(c)
dt = ideal.cfl_dt(c)
for it in range(3): advect(d, opt, dt)
d.synchronize(); t = time.time()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for it in range(n): advect(d, opt, dt)
d.synchronize(); ms = (time.time() - t) / n * 1e3
print("advect ms/call", ms, " nz", nz, " ns per scalar-cell %.4f" % (ms * 1e6 / (nx * ny * nz * 9)))
