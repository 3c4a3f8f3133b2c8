"""advect time vs number of scalars (stream-count sensitivity)"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.grid import grid_t
from icar_amd.domain import domain_t
from icar_amd.advection import advect
from icar_amd.constants import kADV_MPDATA, ADVECTION_ORDER
nx = ny = 512; nz = 40
c = ideal.make_case(nx, ny, nz, hill_height=1000., noise=0.01, n_hydro=1)
d = domain_t(grid_t().set_grid_dimensions(nx, ny, nz, 1, 1)); d.load_case(c)
dt = ideal.cfl_dt(c)
names = ["water_vapor", "cloud_water", "rain_in_air", "snow_in_air", "potential_temperature", "cloud_ice", "graupel_in_air", "ice_number_concentration", "rain_number_concentration"]
for n in (1, 2, 3, 5, 9):
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.advect_vars(names[:n])
    for it in range(2): advect(d, opt, dt)
    d.synchronize(); t = time.time()
    for it in range(5): advect(d, opt, dt)
    d.synchronize(); print(n, "scalars: advect ms/call", (time.time() - t) / 5 * 1e3, " per scalar", (time.time() - t) / 5 * 1e3 / n)
