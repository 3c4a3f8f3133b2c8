"""debug: whole-step CPU chain vs device plain sequence, operator by operator"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.capi import lib, check
from icar_amd.microphysics import mp, mp_init, mp_var_request
from icar_amd.advection import advect, adv_init
from icar_amd.time_step import update_dt
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON
from util import single_image_domain, MEMBER, bits_equal, nbitdiff
from oracle import orc as oracle
oracle.build()
p_, f_ = options_t().mp_options.as_arrays(); oracle.thompson_init(p_, f_)
ADV_ORDER = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]
nx, ny, nz = 96, 64, 20
c = ideal.make_case(nx, ny, nz, hill_height=900.0, noise=0.01, n_hydro=1)
c["water_vapor"] = (c["water_vapor"] * np.float32(1.35)).astype(np.float32)
rng = np.random.default_rng(5)
c["dzdx"] = (0.05 * rng.standard_normal(c["u"].shape)).astype(np.float32)
c["dzdy"] = (0.05 * rng.standard_normal(c["v"].shape)).astype(np.float32)
dq = {"water_vapor": 1e-8, "potential_temperature": 1e-4, "u": 5e-4, "v": -5e-4, "pressure": 1e-3, "w": 2e-6}
dq = {k: (sc * rng.standard_normal(c[k].shape)).astype(np.float32) for k, sc in dq.items()}
forced = [("water_vapor", True), ("potential_temperature", True), ("u", False), ("v", False), ("pressure", False), ("w", False)]
opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"]); opt.parameters.ideal = True
mp_var_request(opt)
d = single_image_domain(c)
check(lib().icar_hip_mpdata_exact(d.ctx, 1), "x")
mp_init(opt, d); adv_init(d, opt)
for k, a in dq.items(): d.set_dqdt(k, a)
f32 = np.float32
s = {k: c[k].copy() for k in ADV_ORDER + ["u", "v", "w", "pressure"]}
dev = dict(MEMBER); dev.update({"u": "u", "v": "v", "w": "w", "pressure": "pressure"})
def cmp(tag):
    bad = [(n, nbitdiff(d.get(dev[n]), s[n])) for n in s if not bits_equal(d.get(dev[n]), s[n])]
    print(tag, "OK" if not bad else bad, flush=True)
    return not bad
dt0 = min(float(f32(0.9) / f32(oracle.max_courant(c["u"], c["v"], c["w"], c["dz_levels"], float(c["dx"])))), 120.0)
end = 6.4 * dt0; t = 0.0; t_mp = None
from icar_amd.constants import ADVECTION_ORDER
for it in range(10):
    if not t < end: break
    dt = update_dt(d, opt)
    if t + dt > end: dt = end - t
    enforce = (end - t) < dt * 2
    dtc = min(float(f32(0.9) / f32(oracle.max_courant(s["u"], s["v"], s["w"], c["dz_levels"], float(c["dx"])))), 120.0)
    print("step", it, "dt", dt, dtc, dt == dtc)
    d.diagnostic_update()
    diag = oracle.diagnostic_update(s["pressure"], s["potential_temperature"], s["u"], s["v"], s["w"], c["dzdx"], c["dzdy"], c["jacobian"])
    print(" diag exner", bits_equal(d.get("exner"), diag["exner"]), "density", bits_equal(d.get("density"), diag["density"]))
    pre = {n: s[n].copy() for n in s}
    mp(d, opt, dt)
    z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
    oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                    s["rain_number"], s["potential_temperature"], diag["exner"], s["pressure"], c["dz_mass"], (float(f32(dt)) if t_mp is None else float(f32(t - t_mp))), *z,
                    1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
    if not cmp(" after mp"):
        import os
        for n in ADV_ORDER:
            g = d.get(dev[n]); w = np.argwhere(g.view(np.int32) != s[n].view(np.int32))
            if len(w): print("  ", n, "first diffs (j,k,i):", w[:4].tolist(), "dev", [float(g[tuple(x)]) for x in w[:4]], "orc", [float(s[n][tuple(x)]) for x in w[:4]])
        w = np.argwhere(d.get(dev["rain"]).view(np.int32) != s["rain"].view(np.int32))
        j, i = int(w[0][0]), int(w[0][2])
        mpdt = (float(f32(dt)) if t_mp is None else float(f32(t - t_mp)))
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez("gpurun_out/th_mismatch_column.npz", j=j, i=i, mp_dt=mpdt, exner=diag["exner"][j, :, i], pressure=pre["pressure"][j, :, i], dz=c["dz_mass"][j, :, i],
                 **{"in_" + n: pre[n][j, :, i] for n in ADV_ORDER}, **{"dev_" + n: d.get(dev[n])[j, :, i] for n in ADV_ORDER}, **{"orc_" + n: s[n][j, :, i] for n in ADV_ORDER})
        print("saved column", j, i, "mp_dt", mpdt); break
    advect(d, opt, dt)
    q = np.stack([s[n] for n in ADV_ORDER]).copy()
    oracle.advect(2, q, s["u"], s["v"], s["w"], diag["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                  c["advection_dz"], c["dz_levels"], float(c["dx"]), float(f32(dt)))
    for m, n in enumerate(ADV_ORDER): s[n] = q[m].copy()
    cmp(" after advect")
    d.apply_forcing(dt, forced)
    for n, fb in forced: oracle.apply_forcing(s[n], dq[n], dt, int(fb), 1, 1, 1, 1)
    cmp(" after forcing")
    if enforce:
        d.enforce_limits([n for n in ADVECTION_ORDER if opt.vars_to_advect.get(n, 0) > 0])
        for n in ADV_ORDER: oracle.enforce_limits(s[n])
        cmp(" after enforce")
    t_mp = t
    d.model_time_seconds += dt; t += dt
