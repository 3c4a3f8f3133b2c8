#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN KERNELS
(oracle/_ref/libicar_ref.so = /root/reference/src/physics/*.f90 compiled unmodified, see
oracle/build_ref.sh).  Only runs in the container that has /root/reference; the .npz files it
writes are data (inputs are regenerated from icar_amd.ideal with the recorded parameters, the
expected outputs are stored) and are committed so that the oracle stays pinned on the GPU box.

The reference keeps module-level arrays sized at first call => one process per case.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# Nt_c, TNO, am_s, rho_g, av_s, bv_s, fv_s, av_g, bv_g, av_i, Ef_si, Ef_rs, Ef_rg, Ef_ri, C_cubes, C_sqrd, mu_r, t_adjust
ALT_MP = [50.e6, 4.0, 0.08, 400.0, 35.0, 0.5, 80.0, 400.0, 0.85, 1800.0, 0.07, 0.9, 0.7, 0.9, 0.4, 0.25, 1.0, 1.0]

CASES = {
    # name: (kind, parameters)
    "adv_upwind_24x20x10": dict(kind="adv", scheme=1, nx=24, ny=20, nz=10, hill=600.0, dens=0, order=2, fct=1, nsteps=3,
                                vars=["water_vapor", "potential_temperature"]),
    "adv_mpdata_24x20x10": dict(kind="adv", scheme=2, nx=24, ny=20, nz=10, hill=600.0, dens=0, order=2, fct=1, nsteps=3,
                                vars=["water_vapor", "potential_temperature", "ice_number"]),
    "adv_mpdata_dens_40x36x12": dict(kind="adv", scheme=2, nx=40, ny=36, nz=12, hill=1000.0, dens=1, order=2, fct=1, nsteps=2,
                                     vars=["water_vapor", "cloud_water"]),
    "adv_mpdata_nofct_40x36x12": dict(kind="adv", scheme=2, nx=40, ny=36, nz=12, hill=1000.0, dens=0, order=2, fct=0, nsteps=2,
                                      vars=["water_vapor"]),
    "adv_mpdata_order1_40x36x12": dict(kind="adv", scheme=2, nx=40, ny=36, nz=12, hill=1000.0, dens=0, order=1, fct=1, nsteps=2,
                                       vars=["water_vapor"]),
    # winds with white noise of 0.5 m/s on u and v, w rebalanced (tests/util.py:roughen_winds): neighbouring Courant numbers differ in
    # sign and size, the limiter works on nearly every face -- the regime in which the all-or-nothing factor next to the ring
    # (adv_mpdata_FCT_core.f90:80-113 with fin = fout = 0) decides cells (round 6); three steps so that the roughened field is advected again
    "adv_mpdata_rough_40x36x12": dict(kind="adv", scheme=2, nx=40, ny=36, nz=12, hill=1000.0, dens=0, order=2, fct=1, nsteps=3, rough=0.5,
                                      vars=["water_vapor", "potential_temperature", "cloud_water"]),
    "adv_mpdata_rough_dens_order3_40x36x12": dict(kind="adv", scheme=2, nx=40, ny=36, nz=12, hill=1000.0, dens=1, order=3, fct=1, nsteps=2, rough=0.5,
                                                  vars=["water_vapor", "potential_temperature"]),
    "adv_mpdata_100x100x30": dict(kind="adv", scheme=2, nx=100, ny=100, nz=30, hill=1000.0, dens=0, order=2, fct=1, nsteps=10,
                                  vars=["water_vapor"], summary_only=1),
    "adv_upwind_100x100x30": dict(kind="adv", scheme=1, nx=100, ny=100, nz=30, hill=1000.0, dens=0, order=2, fct=1, nsteps=10,
                                  vars=["water_vapor"], summary_only=1),
    "mp_simple_40x36x20": dict(kind="mps", nx=40, ny=36, nz=20, hill=1000.0, moist=1.6, cool=0.4, dt=40.0, nsteps=8),
    "mp_simple_snow_30x20x30": dict(kind="mps", nx=30, ny=20, nz=30, hill=500.0, moist=2.5, cool=1.5, dt=60.0, nsteps=8),
    "thompson_warm_24x12x30": dict(kind="th", nx=24, ny=12, nz=30, hill=1000.0, moist=1.6, cool=1.0, dt=40.0, nsteps=12),
    "thompson_cold_20x10x40": dict(kind="th", nx=20, ny=10, nz=40, hill=1000.0, moist=2.0, cool=2.0, dt=60.0, nsteps=25),
    "thompson_longdt_16x8x40": dict(kind="th", nx=16, ny=8, nz=40, hill=1000.0, moist=2.5, cool=3.0, dt=130.0, nsteps=30),
    "thompson_tables": dict(kind="thtab"),
    # a second, non-default mp_options set (opt_types.f90:30-41 order; both efficiency-table flags on): its own table cache
    "thompson_alt_tables": dict(kind="thtab", mp=ALT_MP, flags=[1, 1], cache="/tmp/oracle/run_alt"),
    "thompson_alt_cold_20x10x40": dict(kind="th", nx=20, ny=10, nz=40, hill=1000.0, moist=2.0, cool=2.0, dt=60.0, nsteps=20,
                                       mp=ALT_MP, flags=[1, 1], cache="/tmp/oracle/run_alt"),
}
TH_KEYS = ["water_vapor", "cloud_water", "rain", "cloud_ice", "snow", "graupel", "ice_number", "rain_number", "potential_temperature"]


INPUTS_ADV = ["u", "v", "w", "density", "jacobian", "jacobian_u", "jacobian_v", "jacobian_w", "advection_dz", "dz_levels"]
INPUTS_MPS = ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain", "snow", "dz_mass"]


def summary(a):
    import numpy as np
    a64 = a.astype(np.float64)
    return dict(sum=float(a64.sum()), min=float(a.min()), max=float(a.max()), sumsq=float((a64 * a64).sum()))


def _th_init(ref, p):
    """thompson_init with the case's mp_options; the reference reads whatever *.dat cache it finds in the CWD without checking
    the parameters, so every parameter set has its own directory."""
    import numpy as np
    if "mp" in p:
        ref.thompson_init(np.asarray(p["mp"], np.float32), p["flags"], workdir=p["cache"])
    else:
        ref.thompson_init(workdir=os.environ.get("ICAR_THOMPSON_CACHE", "/tmp/oracle/run"))


def run_case(name):
    import numpy as np
    from oracle import ref
    from icar_amd import ideal
    p = CASES[name]
    if p["kind"] == "adv":
        exact = bool(p.get("summary_only"))
        c = ideal.make_case(p["nx"], p["ny"], p["nz"], hill_height=p["hill"], noise=0.01, n_hydro=1, exact=exact)
        if p.get("rough"):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from util import roughen_winds
            from oracle import orc
            orc.build()
            c = roughen_winds(c, orc, p["rough"])          # (the inputs travel with the fixture: in_u, in_v, in_w)
        dt = ideal.cfl_dt(c)
        q = np.stack([c[n] for n in p["vars"]]).copy()
        ref.advect(p["scheme"], q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"],
                   c["jacobian_w"], c["advection_dz"], c["dz_levels"], float(c["dx"]), dt, advect_density=p["dens"],
                   mpdata_order=p["order"], fct=p["fct"], nsteps=p["nsteps"])
        out = {"dt": np.float64(dt)}
        if p.get("summary_only"):
            out["summary"] = np.array(json.dumps({n: summary(q[m]) for m, n in enumerate(p["vars"])}))
            # a few full k-j planes as spot checks
            out["plane_j50"] = q[:, 50].copy()
        else:
            out["q"] = q
            for n in INPUTS_ADV + p["vars"]:       # small case: the inputs travel with the expected outputs
                out["in_" + n] = c[n]
        # inputs fingerprint so a drift of icar_amd.ideal is detected rather than silently re-baselined
        out["input_sum"] = np.float64(sum(float(c[n].astype(np.float64).sum()) for n in p["vars"]))
    elif p["kind"] == "thtab":
        # the reference's own lookup tables: fingerprints + probed entries (the tables total 85 MB)
        import hashlib
        _th_init(ref, p)
        out = {}
        rng = np.random.default_rng(7)
        for tname in ref.THOMPSON_TABLES:
            t = ref.thompson_table(tname)
            idx = np.sort(rng.choice(t.size, size=min(64, t.size), replace=False))
            out["sha_" + tname] = np.array(hashlib.sha256(t.tobytes()).hexdigest())
            out["idx_" + tname] = idx; out["val_" + tname] = t[idx]; out["sum_" + tname] = np.float64(t.sum())
    elif p["kind"] == "th":
        nx, ny, nz = p["nx"], p["ny"], p["nz"]
        _th_init(ref, p)
        c = ideal.make_case(nx, ny, nz, hill_height=p["hill"], noise=0.01)
        s = {k: c[k].copy() for k in TH_KEYS + ["exner", "pressure", "dz_mass"]}
        s["water_vapor"] = (s["water_vapor"] * np.float32(p["moist"])).astype(np.float32)
        ins = {"in_" + k: s[k].copy() for k in s}
        acc = {k: np.zeros((ny, nx), np.float32) for k in ("rainnc", "rainncv", "snownc", "graupelnc", "sr")}
        for _ in range(p["nsteps"]):
            ref.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                         s["rain_number"], s["potential_temperature"], s["exner"], s["pressure"], s["dz_mass"], p["dt"],
                         acc["rainnc"], acc["rainncv"], acc["snownc"], acc["graupelnc"], acc["sr"],
                         1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
            s["potential_temperature"] -= np.float32(p["cool"])
        out = {k: s[k] for k in TH_KEYS}
        out.update({k: acc[k] for k in ("rainnc", "snownc", "graupelnc")})
        out.update(ins)
    else:
        nx, ny, nz = p["nx"], p["ny"], p["nz"]
        c = ideal.make_case(nx, ny, nz, hill_height=p["hill"], noise=0.01)
        s = {k: c[k].copy() for k in ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water",
                                      "rain", "snow", "dz_mass"]}
        s["water_vapor"] = (s["water_vapor"] * np.float32(p["moist"])).astype(np.float32)
        rain = np.zeros((ny, nx), np.float32); snow = np.zeros((ny, nx), np.float32)
        ins = {"in_" + k: s[k].copy() for k in INPUTS_MPS}
        for _ in range(p["nsteps"]):
            ref.mp_simple(s["pressure"], s["potential_temperature"], s["exner"], s["density"], s["water_vapor"], s["cloud_water"],
                          s["rain"], s["snow"], rain, snow, p["dt"], s["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
            s["potential_temperature"] -= np.float32(p["cool"])
        out = {k: s[k] for k in ["potential_temperature", "water_vapor", "cloud_water", "rain", "snow"]}
        out["rain_acc"] = rain; out["snow_acc"] = snow
        out.update(ins)
        out["input_sum"] = np.float64(float(c["water_vapor"].astype(np.float64).sum()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=np.array(json.dumps(p)), **out)
    print("wrote", name)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for n in CASES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n])
