import sys, types, numpy as np
sys.path.insert(0, ".")
import torch, bench
args = types.SimpleNamespace(nx=512, ny=512, nz=40, hill=1000.0, adv="mpdata", mp="thompson")
d, opt, case, g = bench.build_tile(args, 0, 1, 0)
for _ in range(13): bench.one_step(d, opt)
names = ["water_vapor", "cloud_water_mass", "rain_mass", "snow_mass", "potential_temperature", "cloud_ice_mass", "graupel_mass", "cloud_ice_number", "rain_number"]
tot = 0
for n in names:
    a = d.get(n); nzm = a != 0
    # dilate by 1 in all directions: cell needs work if any neighbour non-zero
    m = nzm.copy()
    for ax in range(3):
        s = m.copy(); s[(slice(1, None) if ax == 0 else slice(None), slice(1, None) if ax == 1 else slice(None), slice(1, None) if ax == 2 else slice(None))] |= m[(slice(None, -1) if ax == 0 else slice(None), slice(None, -1) if ax == 1 else slice(None), slice(None, -1) if ax == 2 else slice(None))]
        s[(slice(None, -1) if ax == 0 else slice(None), slice(None, -1) if ax == 1 else slice(None), slice(None, -1) if ax == 2 else slice(None))] |= m[(slice(1, None) if ax == 0 else slice(None), slice(1, None) if ax == 1 else slice(None), slice(1, None) if ax == 2 else slice(None))]
        m = s
    # per 64-wide row segment (a wave in the fluxes kernel)
    rows = m.reshape(m.shape[0], m.shape[1], -1, 64).any(axis=3)
    print(f"{n:24s} nonzero {nzm.mean():.3f}  needs-work cells {m.mean():.3f}  needs-work waves {rows.mean():.3f}  min nonzero {np.abs(a[nzm]).min() if nzm.any() else 0:.2e}")
    tot += rows.mean()
print("mean needs-work wave fraction over 9 scalars", tot / 9)
