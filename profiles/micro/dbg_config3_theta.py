"""Debug aid (round 6): the MPDATA leg of tests/test_gpu_trajectory.py::test_config3_tile_update_winds_then_substep
(512 x 256 x 40, winds from a random linear-theory LUT) on its own, so that one lease can run it under several builds of mpdata.hip.

  python profiles/micro/dbg_config3_theta.py prepare            -> /tmp/c3_case.npz  (CPU only: oracle winds, Thompson, MPDATA)
  ICAR_HIP_LIB=... python profiles/micro/dbg_config3_theta.py run [tag]   -> per-field error, worst cell (device MPDATA vs the oracle)

The winds and the microphysics of that test are bit-identical on the device, so the device leg starts from the oracle's own
post-Thompson state and winds: the only thing that differs between builds is the fused kernel's arithmetic."""
import sys, os, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from icar_amd import ideal

ADV_ORDER = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]
NPZ = os.environ.get("C3_NPZ", "/tmp/c3_case.npz")
nx, ny, nz = (int(x) for x in os.environ.get("C3_SHAPE", "512,256,40").split(","))


def make_case():
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.5)).astype(np.float32)
    return c


def prepare():
    from oracle import orc, wind_oracle as W
    from icar_amd.options import lt_options_type
    orc.build()
    oracle = orc
    from icar_amd.options import options_t
    th_oracle = orc
    th_oracle.thompson_init(*options_t().mp_options.as_arrays())
    c = make_case()
    dxf = float(c["dx"])
    rng = np.random.default_rng(77)
    ndir, nspd, nnsq = 4, 3, 2
    lt = lt_options_type(buffer=4, n_dir_values=ndir, n_spd_values=nspd, n_nsq_values=nnsq, stability_window_size=3,
                         vert_smooth=2, variable_N=True, smooth_nsq=True, linear_contribution=0.5, linear_update_fraction=1.0)
    zc = (np.cumsum(c["dz_levels"]) - c["dz_levels"] / 2).astype(np.float32)
    z3 = np.ascontiguousarray(c["terrain"][:, None, :] + zc[None, :, None] * np.ones((ny, 1, nx), np.float32), np.float32)
    ulut = (0.5 * rng.standard_normal((ny, nz, nx + 1, nnsq, ndir, nspd))).astype(np.float32)
    vlut = (0.5 * rng.standard_normal((ny + 1, nz, nx, nnsq, ndir, nspd))).astype(np.float32)
    lo, hi = lt.resolved()
    dirv = W.linear_space(lt.dirmin, lt.dirmax, ndir); spdv = W.linear_space(lt.spdmin, lt.spdmax, nspd); nsqv = W.linear_space(lo, hi, nnsq)
    o = dict(variable_N=True, smooth_nsq=True, N_squared=lt.N_squared, max_stability=lt.max_stability, min_stability=lt.min_stability,
             linear_contribution=lt.linear_contribution, linear_update_fraction=lt.linear_update_fraction)
    hyd = tuple(c[k] for k in ("cloud_water", "cloud_ice", "rain", "snow"))
    oracle.set_math_mode(0)
    u, v = c["u"].copy(), c["v"].copy()
    up = np.zeros_like(u); vp = np.zeros_like(v)
    oracle.make_winds_grid_relative(u, v, np.zeros((ny, nx)), np.ones((ny, nx)))
    oracle.spatial_winds(u, v, c["potential_temperature"], c["exner"], z3, c["water_vapor"], hyd, ulut, vlut, up, vp, o, dirv, spdv, nsqv,
                         lt.vert_smooth, lt.stability_window_size)
    w = oracle.balance_uvw(u, v, c["jacobian_u"], c["jacobian_v"], c["jacobian_w"], c["advection_dz"], dxf)
    cw = dict(c); cw["u"], cw["v"], cw["w"] = u, v, w
    dt = float(np.float32(min(ideal.cfl_dt(cw), 60.0)))
    s = {n: c[n].copy() for n in ADV_ORDER}
    z = [np.zeros((ny, nx), np.float32) for _ in range(5)]
    th_oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                       s["rain_number"], s["potential_temperature"], c["exner"], c["pressure"], c["dz_mass"], dt, *z,
                       1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
    q = np.stack([s[n] for n in ADV_ORDER]).copy()
    oracle.advect(2, q, u, v, w, c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                  c["advection_dz"], c["dz_levels"], dxf, dt)
    np.savez(NPZ, u=u, v=v, w=w, dt=dt, **{"in_" + n: s[n] for n in ADV_ORDER}, **{"ref_" + n: q[m] for m, n in enumerate(ADV_ORDER)})
    print("prepared", NPZ, "dt", dt, "max|u - u0|", float(abs(u - c["u"]).max()))


def run(tag):
    from icar_amd.options import options_t
    from icar_amd.advection import advect, adv_init
    from icar_amd.microphysics import mp_var_request
    from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON
    from util import single_image_domain, MEMBER, local_rel_err
    z = np.load(NPZ)
    c = make_case()
    c["u"], c["v"], c["w"] = z["u"], z["v"], z["w"]
    for n in ADV_ORDER: c[n] = z["in_" + n]
    dt = float(z["dt"])
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"])
    mp_var_request(opt)
    d = single_image_domain(c)
    adv_init(d, opt)
    advect(d, opt, dt)
    out = {}
    for n in ADV_ORDER:
        got = d.get(MEMBER[n]); ref = z["ref_" + n]
        e, where = local_rel_err(got, ref)
        out[n] = e
        extra = ""
        if n == "potential_temperature":
            diff = np.abs(got.astype(np.float64) - ref)
            extra = f" cells>1e-6*300: {int((diff > 3e-4).sum())}  >3e-6*300: {int((diff > 9e-4).sum())}  got {got[where]!r} ref {ref[where]!r} in {z['in_' + n][where]!r}"
            np.save(f"/tmp/c3_theta_{tag}.npy", got)
        print(f"[{tag}] {n:22s} {e:.3e} at (j,k,i)={where}{extra}", flush=True)
    d.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "c3_theta.jsonl"), "a") as f:
        f.write(json.dumps({"tag": tag, "err": out}) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "prepare": prepare()
    else: run(sys.argv[2] if len(sys.argv) > 2 else "default")
