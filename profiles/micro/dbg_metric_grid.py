"""debug: metric-grid exact-mode chain, operator by operator"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.capi import lib, check
from icar_amd.microphysics import mp, mp_init, mp_var_request
from icar_amd.advection import advect, adv_init
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON
from util import single_image_domain, MEMBER, bits_equal, nbitdiff
from oracle import orc as oracle
oracle.build()
p_, f_ = options_t().mp_options.as_arrays(); oracle.thompson_init(p_, f_)
ADV_ORDER = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]
nx, ny, nz = 512, 512, 40
c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, seed=1234, n_hydro=1)
c["water_vapor"] = (c["water_vapor"] * np.float32(1.4)).astype(np.float32)
opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"]); opt.parameters.ideal = True
mp_var_request(opt)
d = single_image_domain(c)
check(lib().icar_hip_mpdata_exact(d.ctx, 1), "x")
mp_init(opt, d); adv_init(d, opt)
f32 = np.float32
dt = min(float(f32(0.9) / f32(oracle.max_courant(c["u"], c["v"], c["w"], c["dz_levels"], float(c["dx"])))), 120.0)
s = {n: c[n].copy() for n in ADV_ORDER}
def cmp(tag):
    bad = []
    for n in ADV_ORDER:
        g = d.get(MEMBER[n])
        if not bits_equal(g, s[n]):
            w = np.argwhere(g.view(np.int32) != s[n].view(np.int32))
            bad.append((n, len(w), [(tuple(int(v) for v in x), float(g[tuple(x)]), float(s[n][tuple(x)])) for x in w[:3]]))
    print(tag, "OK" if not bad else bad, flush=True)
    return not bad
for it in range(5):
    pre = {n: s[n].copy() for n in ADV_ORDER}
    mp(d, opt, dt); d.model_time_seconds += dt
    z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
    oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                    s["rain_number"], s["potential_temperature"], c["exner"], c["pressure"], c["dz_mass"], dt, *z,
                    1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
    ok = cmp(f"step {it} after mp")
    if not ok:
        for nn in ADV_ORDER:
            g = d.get(MEMBER[nn]); w = np.argwhere(g.view(np.int32) != s[nn].view(np.int32))
            if len(w): break
        if len(w):
            j, k, i = (int(v) for v in w[0])
            np.savez("gpurun_out/metric_mismatch.npz", j=j, i=i, k=k, mp_dt=dt, exner=c["exner"][j, :, i], pressure=c["pressure"][j, :, i], dz=c["dz_mass"][j, :, i],
                     **{"in_" + n: pre[n][j, :, i] for n in ADV_ORDER}, **{"dev_" + n: d.get(MEMBER[n])[j, :, i] for n in ADV_ORDER}, **{"orc_" + n: s[n][j, :, i] for n in ADV_ORDER})
        break
    pre = {n: s[n].copy() for n in ADV_ORDER}
    advect(d, opt, dt)
    q = np.stack([s[n] for n in ADV_ORDER]).copy()
    oracle.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                  c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
    for m, n in enumerate(ADV_ORDER): s[n] = q[m].copy()
    ok = cmp(f"step {it} after advect")
    if not ok:
        for n in ADV_ORDER:
            g = d.get(MEMBER[n]); w = np.argwhere(g.view(np.int32) != s[n].view(np.int32))
            if len(w):
                j, k, i = (int(v) for v in w[0])
                sl = (slice(max(j - 3, 0), j + 4), slice(max(k - 3, 0), k + 4), slice(max(i - 3, 0), i + 4))
                np.savez("gpurun_out/metric_mismatch_adv.npz", name=n, j=j, k=k, i=i, j0=sl[0].start, k0=sl[1].start, i0=sl[2].start, pre=pre[n][sl], dev=g[sl], orc=s[n][sl],
                         u=c["u"][sl[0], sl[1], slice(sl[2].start, sl[2].stop + 1)], v=c["v"][slice(sl[0].start, sl[0].stop + 1), sl[1], sl[2]], w=c["w"][sl],
                         jaco=c["jacobian"][sl], dz=c["advection_dz"][sl])
                break
        break
