"""What each kernel of the second stream's chain beside the advection costs it (512x512x40): the bench's step with parts of the
chain removed.  Timing experiment only (the removed work is not done elsewhere)."""
import ctypes, os, sys, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench
from icar_amd import capi
from icar_amd.capi import check, lib

ap = argparse.Namespace(nx=512, ny=512, nz=40, hill=1000.0, adv="mpdata", mp="thompson", scaling="strong", gpus=1)
d, opt, case, g = bench.build_tile(ap, 0, 1, 0)
L = lib()


def run(tag, forced, diagnostics, prefetch, n=30):
    d.configure(opt, forced=forced, diagnostics=diagnostics, prefetch_dt=prefetch)
    dt = ctypes.c_double()
    check(L.icar_hip_step_n(d.ctx, 5, ctypes.byref(dt)), "step_n")
    torch.cuda.synchronize()
    L.icar_hip_timing_enable(d.ctx, 1); L.icar_hip_timing_reset(d.ctx)
    import time
    t0 = time.perf_counter()
    check(L.icar_hip_step_n(d.ctx, n, ctypes.byref(dt)), "step_n")
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n * 1e3
    tot = ctypes.c_double(); k = ctypes.c_int()
    L.icar_hip_timing_read(d.ctx, b"advect", ctypes.byref(tot), ctypes.byref(k)); adv = tot.value / max(k.value, 1)
    L.icar_hip_timing_read(d.ctx, b"mp", ctypes.byref(tot), ctypes.byref(k)); mp = tot.value / n
    print(f"{tag:50s} step {el:6.3f}  advect {adv:6.3f}  mp {mp:6.3f}", flush=True)


ring = [("water_vapor", True), ("potential_temperature", True)]
for rep in range(2):
    run("full chain (w_real, forcing u v w p, CFL)", bench.FORCED, True, True)
    run("no whole-field forcing", ring, True, True)
    run("no CFL prefetch", bench.FORCED, True, False)
    run("w_real only", ring, True, False)
    run("nothing beside the advection (no diagnostics)", ring, False, False)
