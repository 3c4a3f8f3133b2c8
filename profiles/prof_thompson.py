import sys, os, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.grid import grid_t
from icar_amd.domain import domain_t
from icar_amd.microphysics import mp, mp_init, mp_var_request
from icar_amd.advection import advect
from icar_amd.constants import kMP_THOMPSON, kADV_MPDATA
nx=ny=int(sys.argv[1]) if len(sys.argv)>1 else 256
nz=40
c=ideal.make_case(nx,ny,nz,hill_height=1000.,noise=0.01,n_hydro=1)
c["water_vapor"]=(c["water_vapor"]*np.float32(1.4)).astype(np.float32)
opt=options_t(); opt.physics.microphysics=kMP_THOMPSON; opt.physics.advection=kADV_MPDATA; mp_var_request(opt)
d=domain_t(grid_t().set_grid_dimensions(nx,ny,nz,1,1)); d.load_case(c); mp_init(opt,d)
dt=60.0
for it in range(6):
    mp(d,opt,dt); d.model_time_seconds+=dt
    if len(sys.argv)>2: advect(d,opt,dt)
d.synchronize(); t=time.time()
for it in range(4):
    mp(d,opt,dt); d.model_time_seconds+=dt
d.synchronize(); print("thompson nx", nx, "ms/call", (time.time()-t)/4*1e3)
