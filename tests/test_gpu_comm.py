"""The transport behind the C ABI (icar_amd/csrc/comm.hip): icar_hip_comm_init / _init_host, icar_hip_halo_send /
_retrieve, icar_hip_co_min, icar_hip_update_dt.

The GPU box has ONE GPU and RCCL refuses two ranks on one device, so RCCL itself is exercised with a one-rank communicator
whose north and south neighbour is the image itself: the same ncclSend / ncclRecv group, ncclAllReduce and stream ordering an
8-rank run issues, checked against the transport-free periodic wrap (ICAR_NEIGHBOR_SELF) bit for bit.  Several images on the
one GPU go through the host-staged transport (tests/test_gpu_multirank.py, tests/test_gpu_fortran_host.py)."""
import ctypes
import os
import subprocess
import sys
import textwrap
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_self_ring_equals_periodic_wrap_and_device_side_update_dt():
    """Runs in a child process: librccl is 573 MB and is only loaded by icar_hip_comm_init with a unique id."""
    code = textwrap.dedent("""
        import ctypes, os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        from icar_amd import ideal, capi
        from icar_amd.capi import lib, check, NEIGHBOR_NONE, NEIGHBOR_SELF, COMM_RCCL, COMM_LOCAL
        from icar_amd.options import options_t
        from icar_amd.time_step import compute_dt, update_dt
        from util import single_image_domain
        L = lib()
        c = ideal.make_case(70, 41, 12, hill_height=900.0, noise=0.02)
        rng = np.random.default_rng(5)
        fields = [0, 4, 2]                                   # water_vapor, potential_temperature, rain
        ids = (ctypes.c_int * 3)(*fields)
        def fresh():
            d = single_image_domain(c)
            for f in fields:
                d.set(f, rng.standard_normal(d.shape(f)).astype(np.float32))
            return d
        rng = np.random.default_rng(5); a = fresh()
        rng = np.random.default_rng(5); b = fresh()
        # a: RCCL, one rank, my north and south neighbour is myself.  b: the periodic wrap without transport.
        uid = ctypes.create_string_buffer(128)
        check(L.icar_hip_comm_unique_id(uid), "unique_id")
        assert any(uid.raw)
        check(L.icar_hip_comm_init(a.ctx, 1, 0, uid.raw, (ctypes.c_int * 4)(0, 0, NEIGHBOR_NONE, NEIGHBOR_NONE)), "comm_init rccl")
        check(L.icar_hip_comm_init(b.ctx, 1, 0, None, (ctypes.c_int * 4)(NEIGHBOR_SELF, NEIGHBOR_SELF, NEIGHBOR_NONE, NEIGHBOR_NONE)), "comm_init local")
        assert L.icar_hip_comm_kind(a.ctx) == COMM_RCCL and L.icar_hip_comm_kind(b.ctx) == COMM_LOCAL
        for rep in range(3):                                 # several exchanges: buffers are reused, stream order is the only sync
            for d in (a, b):
                check(L.icar_hip_halo_send(d.ctx, 1, ids, 3), "halo_send")
                check(L.icar_hip_halo_retrieve(d.ctx, 1, ids, 3), "halo_retrieve")
            for f in fields:
                x, y = a.get(f), b.get(f)
                assert np.array_equal(x, y), (rep, f)
                assert np.array_equal(x[0], x[-2]) and np.array_equal(x[-1], x[1])       # south halo row <- north edge, north halo row <- south edge
                a.set(f, (x * np.float32(1.5)).astype(np.float32)); b.set(f, (y * np.float32(1.5)).astype(np.float32))
        # a second send without a retrieve is refused
        check(L.icar_hip_halo_send(a.ctx, 1, ids, 3), "halo_send")
        assert L.icar_hip_halo_send(a.ctx, 1, ids, 3) != 0
        check(L.icar_hip_halo_retrieve(a.ctx, 1, ids, 3), "halo_retrieve")
        # co_min / co_max through ncclAllReduce
        v = ctypes.c_double(37.25); check(L.icar_hip_co_min(a.ctx, ctypes.byref(v)), "co_min"); assert v.value == 37.25
        v = ctypes.c_double(-2.5); check(L.icar_hip_co_max(a.ctx, ctypes.byref(v)), "co_max"); assert v.value == -2.5
        # update_dt: with RCCL and cfl_strictness 3 / 4 the tile maximum is all-reduced on the device; equal to the host-combined
        # compute_dt bit for bit.  (The other settings compare m/s with a Courant number and stop with "time step too small"
        # on any realistic wind, here as in the reference: tests/test_gpu_step_rows.py.)
        for strict in (3, 4):
            opt = options_t(); opt.parameters.cfl_strictness = strict
            opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"])
            want = min(compute_dt(a, opt), 120.0)
            assert update_dt(a, opt) == want == update_dt(b, opt), strict
        check(L.icar_hip_comm_destroy(a.ctx), "comm_destroy")
        a.close(); b.close()
        # the whole sub-step loop with the transfer inside it: icar_hip_step_n issues the strips, the pack, the RCCL send / recv group
        # and the wind setup on the second stream beside the interior microphysics, joins, unpacks, advects.  Image a exchanges its
        # north / south edges with itself THROUGH RCCL, image b wraps them without transport: every field bit for bit after 4 sub-steps.
        from icar_amd.time_step import step_n
        from icar_amd.microphysics import mp_init, mp_var_request
        from icar_amd.advection import adv_init
        from icar_amd.constants import kADV_MPDATA, kMP_THOMPSON, ADVECTION_ORDER
        c2 = ideal.make_case(64, 40, 20, hill_height=900.0, noise=0.02, n_hydro=1)
        c2["water_vapor"] = (c2["water_vapor"] * np.float32(1.8)).astype(np.float32)
        opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.physics.microphysics = kMP_THOMPSON
        opt.parameters.dz_levels = c2["dz_levels"]; opt.parameters.dx = float(c2["dx"])
        mp_var_request(opt)
        outs = []
        for kind in ("rccl", "local"):
            d = single_image_domain(c2)
            d.exchange_vars = [n for n in ADVECTION_ORDER if opt.vars_to_advect.get(n, 0) > 0]
            class _Halo: halo = 1
            d.comm = _Halo()                      # domain_t.halo_send / configure only ask the comm for the halo width
            if kind == "rccl":
                check(L.icar_hip_comm_unique_id(uid), "unique_id")
                check(L.icar_hip_comm_init(d.ctx, 1, 0, uid.raw, (ctypes.c_int * 4)(0, 0, NEIGHBOR_NONE, NEIGHBOR_NONE)), "comm_init rccl")
            else:
                check(L.icar_hip_comm_init(d.ctx, 1, 0, None, (ctypes.c_int * 4)(NEIGHBOR_SELF, NEIGHBOR_SELF, NEIGHBOR_NONE, NEIGHBOR_NONE)), "comm_init local")
            mp_init(opt, d); adv_init(d, opt)
            step_n(d, 4, opt)
            outs.append({n: d.get(n) for n in ("water_vapor", "cloud_water_mass", "rain_mass", "snow_mass", "potential_temperature", "cloud_ice_mass",
                                               "graupel_mass", "cloud_ice_number", "rain_number", "accumulated_precipitation")})
            d.close()
        for n in outs[0]:
            assert np.array_equal(outs[0][n], outs[1][n]), n
        assert float(outs[0]["cloud_water_mass"].max()) > 1e-6
        x = outs[0]["water_vapor"]
        assert np.array_equal(x[0], x[0]) and not np.array_equal(x[1], c2["water_vapor"][1])
        print("RCCL_COMM_OK")
    """) % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_COMM_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_comm_init_argument_checks():
    from icar_amd import ideal
    from icar_amd.capi import lib, NEIGHBOR_NONE
    from util import single_image_domain
    L = lib()
    d = single_image_domain(ideal.make_case(20, 16, 6))
    none4 = (ctypes.c_int * 4)(*[NEIGHBOR_NONE] * 4)
    assert L.icar_hip_comm_init(d.ctx, 2, 0, None, none4) != 0                       # several images need a unique id
    assert b"unique id" in L.icar_hip_last_error()
    assert L.icar_hip_comm_init(d.ctx, 1, 0, None, (ctypes.c_int * 4)(0, NEIGHBOR_NONE, NEIGHBOR_NONE, NEIGHBOR_NONE)) != 0
    assert L.icar_hip_comm_init(d.ctx, 1, 0, None, (ctypes.c_int * 4)(5, NEIGHBOR_NONE, NEIGHBOR_NONE, NEIGHBOR_NONE)) != 0
    assert L.icar_hip_comm_init(d.ctx, 1, 3, None, none4) != 0
    assert L.icar_hip_comm_init(d.ctx, 1, 0, None, none4) == 0
    ids = (ctypes.c_int * 1)(0)
    assert L.icar_hip_halo_retrieve(d.ctx, 1, ids, 1) != 0                            # no send before
    v = ctypes.c_double(3.0)
    assert L.icar_hip_co_min(d.ctx, ctypes.byref(v)) == 0 and v.value == 3.0
    d.close()
