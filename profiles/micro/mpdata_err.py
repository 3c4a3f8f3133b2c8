"""Deviation of the fused MPDATA kernel from the CPU oracle (bit-exact restatement of the reference): per field the number
of cells that differ, max|d| / max|field| and max over cells of |d| / (max |ref| within 2 cells) -- the local-scale
relative error the parity tests bound by 1e-5.  usage: python profiles/micro/mpdata_err.py [nx ny nz nsteps dens fct order]"""
import sys, json
import numpy as np
from scipy.ndimage import maximum_filter
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.advection import advect
from icar_amd.constants import kADV_MPDATA
from util import SCALARS, MEMBER, KVAR, single_image_domain, adv_args
from oracle import orc
orc.build()
a = sys.argv[1:]
nx, ny, nz, nsteps = (int(a[i]) if len(a) > i else v for i, v in enumerate((70, 37, 12, 2)))
dens = bool(int(a[4])) if len(a) > 4 else False
fct = bool(int(a[5])) if len(a) > 5 else True
order = int(a[6]) if len(a) > 6 else 2
import os
names = SCALARS[:int(os.environ.get("NSCAL", "9"))]
c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
dt = ideal.cfl_dt(c)
q = np.stack([c[n] for n in names]).copy()
orc.advect(kADV_MPDATA, q, *adv_args(c), dt, advect_density=dens, mpdata_order=min(order, 2), fct=fct, nsteps=nsteps)
d = single_image_domain(c)
opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.parameters.advect_density = dens
opt.adv_options.mpdata_order = order; opt.adv_options.flux_corrected_transport = fct
opt.advect_vars([KVAR[n] for n in names])
for _ in range(nsteps):
    advect(d, opt, dt)
res = {}
for m, n in enumerate(names):
    got = d.get(MEMBER[n]).astype(np.float64); ref = q[m].astype(np.float64)
    diff = np.abs(got - ref)
    scale = maximum_filter(np.abs(ref), size=5, mode="nearest")
    rel = diff / np.maximum(scale, 1e-300)
    rel[scale == 0] = np.where(diff[scale == 0] == 0, 0.0, np.inf)
    w = np.unravel_index(np.argmax(rel), rel.shape)
    res[n] = dict(ndiff=int((got != ref).sum()), of=int(ref.size), max_over_fieldmax=float(diff.max() / max(np.abs(ref).max(), 1e-300)),
                  max_local_rel=float(rel.max()), at=[int(x) for x in w], nonfinite=int((~np.isfinite(got)).sum()))
    print(n, res[n], flush=True)
d.close()
print("WORST local rel", max(r["max_local_rel"] for r in res.values()))
