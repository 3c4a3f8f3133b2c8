"""Debug aid: one MPDATA case on the device against the oracle, with the error map summarised (not a test)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.advection import advect
from icar_amd.constants import kADV_MPDATA
from util import MEMBER, KVAR, single_image_domain, adv_args, local_rel_err
from oracle import orc
orc.build()
nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (70, 37, 12)
order = int(sys.argv[4]) if len(sys.argv) > 4 else 2
fct = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
names = ["water_vapor", "cloud_water", "potential_temperature"]
c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
dt = ideal.cfl_dt(c)
q = np.stack([c[n] for n in names]).copy()
orc.advect(kADV_MPDATA, q, *adv_args(c), dt, advect_density=False, mpdata_order=order, fct=fct, nsteps=1)
d = single_image_domain(c)
opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.adv_options.mpdata_order = order; opt.adv_options.flux_corrected_transport = fct
opt.advect_vars([KVAR[n] for n in names])
advect(d, opt, dt)
for m, n in enumerate(names):
    got = d.get(MEMBER[n]); ref = q[m]
    e = np.abs(got.astype(np.float64) - ref) / max(float(np.abs(ref).max()), 1e-300)
    j, k, i = np.unravel_index(np.argmax(e), e.shape)
    print(n, "max err/max", e.max(), "at (j,k,i)", (j, k, i), "nan", int(np.isnan(got).sum()), "cells>1e-5:", int((e > 1e-5).sum()), flush=True)
    if (e > 1e-5).any():
        bad = np.argwhere(e > 1e-5)
        print("   bad j:", np.unique(bad[:, 0])[:20], " k:", np.unique(bad[:, 1])[:20], " i:", np.unique(bad[:, 2])[:30])
d.close()
