"""hipGraph replay of the sub-step against the eager launches (icar_hip_substep_graph_probe): timing per tile size, and the state
after the probe compared bit for bit with the same number of eager sub-steps.  usage: graph_probe.py [pairs]"""
import sys, ctypes, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.capi import lib, check
from icar_amd.microphysics import mp_init, mp_var_request
from icar_amd.advection import adv_init, adv_var_request
from icar_amd.time_step import substep, update_dt
from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON, ADVECTION_ORDER
from icar_amd.grid import grid_t
from icar_amd.domain import domain_t
from icar_amd.halo import HaloComm
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
FORCED = [("water_vapor", True), ("potential_temperature", True), ("u", False), ("v", False), ("pressure", False), ("w", False)]
for nx, ny in ((258, 130), (258, 258), (512, 512)):
    nz = 40
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, seed=1234, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.4)).astype(np.float32)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
    opt.parameters.ideal = True; opt.parameters.dx = float(c["dx"]); opt.parameters.dz_levels = c["dz_levels"]
    mp_var_request(opt); adv_var_request(opt)
    def fresh():
        g = grid_t().set_grid_dimensions(nx, ny, nz, 1, 1)
        d = domain_t(g, device=0, dx=float(c["dx"]), image=1, comm=HaloComm(g, 1, loopback=True))
        d.load_case(c)
        d.exchange_vars = [n for n in ADVECTION_ORDER if opt.vars_to_advect.get(n, 0) > 0]
        mp_init(opt, d); adv_init(d, opt)
        for n in FORCED: d.set_dqdt(n[0], np.zeros(d.shape(d.fid(n[0])), np.float32))
        d.set("dzdx", np.zeros(d.shape(d.fid("dzdx")), np.float32)); d.set("dzdy", np.zeros(d.shape(d.fid("dzdy")), np.float32))
        d.configure(opt, forced=FORCED, diagnostics=True, prefetch_dt=False)
        return d
    a = fresh()
    dt = update_dt(a, opt)
    a.configure(opt, forced=FORCED, diagnostics=True, prefetch_dt=False)
    me, mg = ctypes.c_double(), ctypes.c_double()
    check(lib().icar_hip_substep_graph_probe(a.ctx, dt, pairs, ctypes.byref(me), ctypes.byref(mg)), "probe")
    b = fresh()
    for _ in range(2 + 4 * pairs):
        check(lib().icar_hip_substep(b.ctx, dt, 0), "substep"); b.model_time_seconds += dt
    same = all(np.array_equal(a.get(n).view(np.int32), b.get(n).view(np.int32)) for n in ("water_vapor", "cloud_water_mass", "rain_mass", "potential_temperature", "cloud_ice_number", "u", "w_real", "density"))
    print(f"{nx}x{ny}x{nz}: eager {me.value / (2 * pairs):.4f} ms/sub-step, graph replay {mg.value / (2 * pairs):.4f} ms/sub-step ({pairs} replays of 2), state after the probe == eager: {same}", flush=True)
    a.close(); b.close()
