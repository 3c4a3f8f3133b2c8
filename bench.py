#!/usr/bin/env python
"""bench.py -- grid-cell updates/s of ICAR's per-timestep 3-D grid update on MI355X.

A "step" is one pass of the hot path of time_step.f90:440-551 over the tile:
    update_dt (CFL reduction + co_min) -> diagnostic_update -> mp(halo=1) -> halo_send -> mp(subset=1)
    -> halo_retrieve -> advect -> apply_forcing
on the synthetic ideal case of SURVEY.md 8(d): a 512x512x40 grid, MPDATA order 2 + FCT,
Thompson microphysics (9 advected scalars) -- or mp_simple (5 scalars) with --mp simple.
Inputs are resident in HBM before the timed region.  Scaling is STRONG by default, as north_star asks ("cell-updates/sec on
a synthetic 512x512x40 grid reported at 1, 2, 4 and 8 GPUs"): the global grid is fixed and decomposed over the N images exactly
like grid_obj.f90:39-255 (BASELINE.json configs[2] is this grid as 2x2).  --scaling weak keeps a 512x512x40 tile per GPU
instead (the global domain grows with the image grid).  One call per sub-step into the library (icar_hip_update_dt +
icar_hip_substep): the launches, the second stream and the RCCL halo exchange are issued from C.

Prints ONE JSON line (rank 0) with the driver's contract plus "roofline" and "cpu_baseline".
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 measured copy


def build_tile(args, rank, world, device):
    from icar_amd import ideal
    from icar_amd.grid import grid_t, domain_decomposition
    from icar_amd.domain import domain_t
    from icar_amd.options import options_t
    from icar_amd.halo import HaloComm
    from icar_amd.microphysics import mp_var_request, mp_init
    from icar_amd.advection import adv_var_request, adv_init
    from icar_amd.constants import kADV_MPDATA, kADV_UPWIND, kMP_THOMPSON, kMP_SB04, kMP_WSM3, kMP_WSM6, KVARS, ADVECTION_ORDER

    xs, ys = domain_decomposition(args.nx, args.ny, world) if world > 1 else (1, 1)
    if args.scaling == "strong":
        # the global nx x ny x nz grid is fixed; every image cuts its tile (halos included) out of the same case
        gnx, gny = args.nx, args.ny
        g = grid_t().set_grid_dimensions(gnx, gny, args.nz, world, rank + 1)
        whole = ideal.make_case(gnx, gny, args.nz, hill_height=args.hill, noise=0.01, seed=1234, n_hydro=1)
        # moisten so that the microphysics is active in (nearly) every column
        whole["water_vapor"] = (whole["water_vapor"] * np.float32(1.4)).astype(np.float32)
        case = ideal.cut_tile(whole, g) if world > 1 else whole
        del whole
    else:
        # weak scaling: the global domain grows with the image grid so each tile keeps nx x ny owned cells
        gnx, gny = args.nx * xs, args.ny * ys
        g = grid_t().set_grid_dimensions(gnx, gny, args.nz, world, rank + 1)
        tnx, tny = g.ime - g.ims + 1, g.jme - g.jms + 1
        case = ideal.make_case(tnx, tny, args.nz, hill_height=args.hill, noise=0.01, seed=1234 + rank, n_hydro=1)
        case["water_vapor"] = (case["water_vapor"] * np.float32(1.4)).astype(np.float32)
    opt = options_t()
    opt.physics.advection = kADV_UPWIND if args.adv == "upwind" else kADV_MPDATA
    opt.physics.microphysics = {"thompson": kMP_THOMPSON, "simple": kMP_SB04, "wsm3": kMP_WSM3, "wsm6": kMP_WSM6, "none": 0}[args.mp]
    opt.parameters.ideal = True
    opt.parameters.dx = float(case["dx"])
    opt.parameters.dz_levels = case["dz_levels"]
    mp_var_request(opt); adv_var_request(opt)
    comm = HaloComm(g, rank + 1, loopback=(world == 1))
    d = domain_t(g, device=device, dx=float(case["dx"]), image=rank + 1, comm=comm)      # comm.attach: icar_hip_comm_init (RCCL)
    d.load_case(case)
    d.exchange_vars = [n for n in ADVECTION_ORDER if opt.vars_to_advect.get(n, 0) > 0]
    mp_init(opt, d); adv_init(d, opt)
    # forcing tendencies (domain%apply_forcing): boundary relaxation of theta/qv, whole-field u,v,w,p updates.
    # Zero tendencies keep the synthetic state steady; the kernels run and move the same bytes regardless.
    for n in FORCED:
        d.set_dqdt(n[0], np.zeros(d.shape(d.fid(n[0])), np.float32))
    d.set("dzdx", np.zeros(d.shape(d.fid("dzdx")), np.float32)); d.set("dzdy", np.zeros(d.shape(d.fid("dzdy")), np.float32))
    return d, opt, case, g


# the kernels one advect() call launches, per scheme (the roofline's `kernel` label; also the key of profiles/advect_traffic.json)
ADVECT_KERNELS = {"mpdata": "k_mpdata_fused", "upwind": "k_upwind_pass"}
SETUP_KERNELS = ("k_setup_winds", "k_mpdata_coef")      # launched once per step for the advect() call (timer group "winds")
# bumped whenever the advection kernels change what they read or write: profiles/advect_traffic.json (PMC passes) belongs to one
KERNEL_GENERATION = "r06c: r05's branch-free paired steady steps + a donor-cell pass bit-identical to the reference (exact quotients from v_rcp + Newton); 11 coefficient arrays, the final update's reciprocals from the donor-cell pass (registers / LDS): 81 loads per step (k_mpdata_fused + k_mpdata_coef)"

FORCED = [("water_vapor", True), ("potential_temperature", True), ("u", False), ("v", False), ("pressure", False), ("w", False)]


def run_steps(d, opt, n):
    """n steps in ONE library call (icar_hip_step_n): update_dt -> substep -> clock += dt, nothing of the host in between.
    update_dt: the CFL reduction (prefetched beside the last advection) + co_min over RCCL.  substep: diagnostic_update ->
    mp(halo=1) -> halo_send -> mp(subset=1) -> halo_retrieve -> advect -> apply_forcing at EVERY world size: the interior on the
    main stream, the strips + pack (+ RCCL send / recv) + the wind setup beside it on the second stream; the w_real diagnostic,
    the whole-field forcing of u, v, w, p and the next CFL reduction beside the advection.  With one image the edges wrap around
    to the tile itself (ICAR_NEIGHBOR_SELF: same pack / unpack kernels, no transport), so the N=1 line times the launches every
    rank of an N>1 run pays."""
    from icar_amd.time_step import step_n
    return step_n(d, n, opt, forced=FORCED)


def cpu_reference(args, nscalars):
    """The UNMODIFIED reference kernels (adv_mpdata.f90 + mp_thompson.f90 compiled by oracle/build_ref.sh, flang -O2, one core)
    on one step of this workload -- RECORDED, not run here: the compiled reference can only be rebuilt where /root/reference
    exists, so it is timed in the build container (profiles/measure_cpu_reference.py -> profiles/cpu_reference.json, with the CPU it
    ran on) and attached when the record is of this configuration.  Informational, next to cpu_baseline, which IS timed live on
    this box's host cores (the OpenMP restatement that is bit-identical to those kernels)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference.json")))
        want = {"nx": args.nx, "ny": args.ny, "nz": args.nz, "adv": args.adv, "mp": args.mp, "nscalars": nscalars}
        if rec.get("config") != want:
            return None
        return {k: rec[k] for k in ("value", "unit", "cores", "kind", "where", "sample", "recorded")}
    except Exception:
        return None


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max or v1
    cfs_quota).  omp_get_max_threads() reports the host's logical CPUs even inside a quota'd container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(args, nscalars):
    """The CPU restatement (oracle/, bit-identical to the compiled reference kernels) timed on this
    box's host cores on a bounded sample of the same workload."""
    try:
        from oracle import orc
        from icar_amd import ideal
        orc.build()
        if "OMP_NUM_THREADS" not in os.environ:
            orc.set_num_threads(usable_cpus())
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "grid-cell updates/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    nx, ny, nz = args.nx, args.ny, args.nz             # the SAME tile size as the GPU line
    c = ideal.make_case(nx, ny, nz, hill_height=args.hill, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.4)).astype(np.float32)
    dt = min(ideal.cfl_dt(c), 120.0)
    names = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel",
             "ice_number", "rain_number"][:nscalars]
    q = np.stack([c[n] for n in names]).copy()
    rain = np.zeros((ny, nx), np.float32); snow = np.zeros((ny, nx), np.float32)
    s = {k: c[k].copy() for k in ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water",
                                  "rain", "snow", "dz_mass"]}
    scheme = 1 if args.adv == "upwind" else 2
    th = None
    if args.mp == "thompson":
        from icar_amd.options import options_t
        p, f = options_t().mp_options.as_arrays()
        orc.thompson_init(p, f)
        th = {k: c[k].copy() for k in ["cloud_ice", "graupel", "ice_number", "rain_number"]}
        z2 = lambda: np.zeros((ny, nx), np.float32)
        th.update(rainnc=z2(), rainncv=z2(), snownc=z2(), graupelnc=z2(), sr=z2())

    if args.mp == "wsm3":
        orc.wsm3_init()
        w3args = np.array([dt, 9.81, 1012.0, 4 * np.float32(461.6), 287.058, 461.5, 273.15, np.float32(461.5) / np.float32(287.058) - np.float32(1),
                           np.float32(287.058) / np.float32(461.5), 1e-15, 2.85e6, 2.5e6, 3.5e5, 1.28, 1000.0, 4190.0, 2106.0, 610.78], np.float32)
        w3tmp = [np.zeros((ny, nx), np.float32) for _ in range(3)]
    if args.mp == "wsm6":
        orc.wsm6_init()
        w3args = np.array([dt, 9.81, 1012.0, 4 * np.float32(461.6), 287.058, 461.5, 273.15, np.float32(461.5) / np.float32(287.058) - np.float32(1),
                           np.float32(287.058) / np.float32(461.5), 1e-15, 2.85e6, 2.5e6, 3.5e5, 1.28, 1000.0, 4190.0, 2106.0, 610.78], np.float32)
        w6 = {k: c[k].copy() for k in ["cloud_ice", "graupel"]}
        w6.update(sr=np.zeros((ny, nx), np.float32), graupel_acc=np.zeros((ny, nx), np.float32))

    def step():
        if args.mp == "thompson":
            orc.thompson(s["water_vapor"], s["cloud_water"], s["rain"], th["cloud_ice"], s["snow"], th["graupel"], th["ice_number"],
                         th["rain_number"], s["potential_temperature"], s["exner"], s["pressure"], s["dz_mass"], dt,
                         th["rainnc"], th["rainncv"], th["snownc"], th["graupelnc"], th["sr"], 1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
        elif args.mp == "simple":
            orc.mp_simple(s["pressure"], s["potential_temperature"], s["exner"], s["density"], s["water_vapor"],
                          s["cloud_water"], s["rain"], s["snow"], rain, snow, dt, s["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
        elif args.mp == "wsm3":
            orc.wsm3(s["potential_temperature"], s["water_vapor"], s["cloud_water"], s["rain"], c["w"], s["density"], s["exner"], s["pressure"],
                     s["dz_mass"], w3args, rain, w3tmp[0], snow, w3tmp[1], w3tmp[2], 2, nx - 1, 2, ny - 1, 1, nz)
        elif args.mp == "wsm6":
            orc.wsm6(s["potential_temperature"], s["water_vapor"], s["cloud_water"], s["rain"], w6["cloud_ice"], s["snow"], w6["graupel"],
                     s["density"], s["exner"], s["pressure"], s["dz_mass"], w3args, rain, w6["sr"], snow, w6["graupel_acc"], 2, nx - 1, 2, ny - 1, 1, nz)
        orc.advect(scheme, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"],
                   c["jacobian_w"], c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)

    step()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 10.0 or n < 2:
        step(); n += 1
    el = time.perf_counter() - t0
    cells = (nx - 2) * (ny - 2) * nz * n
    mpname = {"thompson": "Thompson", "simple": "mp_simple", "wsm3": "WSM3", "wsm6": "WSM6", "none": "no microphysics"}[args.mp]
    return {"value": cells / el, "unit": "grid-cell updates/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{n} steps of {nx}x{ny}x{nz}, {args.adv} advection of {nscalars} scalars + {mpname}, "
                      f"oracle/*.c CPU restatement (bit-identical to the reference kernels) with OpenMP on {orc.num_threads()} threads, {el:.1f} s"}



def probe_traffic(args):
    """HBM bytes one advect() call moves, measured now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
    pass, MI355X_MICROARCH.md) over a 3-step child run of this same configuration, counters only (--kernel-trace + --pmc, no
    other trace domain).  FETCH_SIZE is doubled (gfx950's rocprofv3 counts 128-B requests as 64 B).  Returns
    (bytes, read, write) or None when rocprofv3 is missing, times out or finds no dispatch of the kernel."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-traffic-probe",
             "--nx", str(args.nx), "--ny", str(args.ny), "--nz", str(args.nz), "--adv", args.adv, "--mp", args.mp, "--hill", str(args.hill)]
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="icar_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--"] + child,
                           cwd="/tmp", env=env, timeout=90, capture_output=True)      # (a child takes ~12 s; a stuck profiler costs the line 90 s, not the run)
            vals = []; setup = {k: [] for k in SETUP_KERNELS}
            for fn in glob.glob(os.path.join(out, "**", "p_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r["Counter_Name"] != ctr:
                        continue
                    if ADVECT_KERNELS[args.adv] in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
                    for k in SETUP_KERNELS:
                        if k in r["Kernel_Name"]:
                            setup[k].append(float(r["Counter_Value"]))
            if not vals:
                return None
            got[ctr] = sum(vals) / len(vals)
            # the once-per-step setup of the advection (Courant winds; the scalar-independent MPDATA coefficients): mean per dispatch each
            got[ctr + "_setup"] = sum(sum(v) / len(v) for k, v in setup.items() if v and (args.adv == "mpdata" or k == "k_setup_winds"))
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    rd, wr = 2.0 * 1024.0 * got["FETCH_SIZE"], 1024.0 * got["WRITE_SIZE"]          # KiB units; x2: see above
    setup_bytes = 2.0 * 1024.0 * got["FETCH_SIZE_setup"] + 1024.0 * got["WRITE_SIZE_setup"]
    return rd + wr, rd, wr, setup_bytes


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU, RCCL) under
    torch.distributed.run on 127.0.0.1 and pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--ny", type=int, default=512)
    ap.add_argument("--nz", type=int, default=40)
    ap.add_argument("--hill", type=float, default=1000.0)
    ap.add_argument("--adv", default="mpdata", choices=["mpdata", "upwind"])
    ap.add_argument("--mp", default=os.environ.get("ICAR_BENCH_MP", "thompson"), choices=["thompson", "simple", "wsm3", "wsm6", "none"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default): the nx x ny x nz grid is global and split over the GPUs; weak: it is the tile of every GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-later-window", action="store_true", help="skip the second (untimed-region) window 100 steps later")
    ap.add_argument("--no-traffic-probe", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--no-kernel-timers", action="store_true", help="profiling: leave the library's per-group HIP-event timers off (roofline.avg_ms is then 0)")
    ap.add_argument("--mpdata-exact", action="store_true",
                    help="advect with icar_hip_mpdata_exact(ctx, 1): MPDATA in the reference's operation order, bit-identical to the CPU reference")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with torch.distributed.run "
                         f"--nproc-per-node N, or run `python bench.py --gpus N` alone and it spawns them")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # ICAR_BENCH_BACKEND=gloo lets several ranks share one GPU (the library's host-staged transport, icar_hip_comm_init_host):
    # a functional check of the N>1 path on a 1-GPU box, never a performance number.  The driver's launches use RCCL.
    backend = os.environ.get("ICAR_BENCH_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    red_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1 or "RANK" in os.environ:           # under a launcher even one rank joins the process group (and gets the RCCL communicator)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: --gpus {world} over RCCL needs {world} visible GPUs, this box has "
                                 f"{torch.cuda.device_count()} (ICAR_BENCH_BACKEND=gloo shares one GPU as a functional check)")
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    in_group = dist.is_initialized()

    from icar_amd import capi
    d, opt, case, g = build_tile(args, rank, world, dev_index)
    lib = capi.lib()
    if args.mpdata_exact:
        capi.check(lib.icar_hip_mpdata_exact(d.ctx, 1), "icar_hip_mpdata_exact")
    kind = int(lib.icar_hip_comm_kind(d.ctx))
    nscal = sum(1 for v in opt.vars_to_advect.values() if v > 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Before anything is timed, the transport proves itself (exchangeable_obj.f90:138-356): the communicator must connect as many
    # images as the launcher started, one exchange of a rank-stamped field must put every neighbour's stamp into the halo cells
    # facing it (checked on the device), and a run with a GPU per rank must not have degraded to the host-staged transport.
    ranks_seen, halo_check = 1, "not applicable (one image, no process group)"
    if in_group:
        nr = ctypes.c_int(0); nbad = ctypes.c_int(-1)
        capi.check(lib.icar_hip_comm_ranks(d.ctx, ctypes.byref(nr)), "icar_hip_comm_ranks")
        capi.check(lib.icar_hip_halo_selfcheck(d.ctx, 1, ctypes.byref(nbad)), "icar_hip_halo_selfcheck")
        ranks_seen = int(nr.value)
        bad_t = torch.tensor([float(nbad.value), float(ranks_seen != world)], dtype=torch.float64, device=red_device)
        if world > 1:
            dist.all_reduce(bad_t)
        halo_check = "ok" if bad_t[0].item() == 0 else f"FAILED: {int(bad_t[0].item())} halo cells do not carry their neighbour's stamp"
        degraded = backend == "nccl" and kind != capi.COMM_RCCL
        if bad_t[0].item() != 0 or bad_t[1].item() != 0 or degraded:
            if rank == 0:
                print(json.dumps({"error": "transport self-check failed", "ranks_seen": ranks_seen, "world": world, "halo_check": halo_check,
                                  "backend_kind": kind, "degraded_to_host_staged": bool(degraded)}), flush=True)
            d.close()
            raise SystemExit(3)

    if args.warmup > 0:
        run_steps(d, opt, args.warmup)
    barrier()
    # Inside the timed region only the advection is bracketed by HIP events (roofline.avg_ms: two timestamped packets per step on
    # the stream it is launched on); the other groups' timers cost ~5 us each and a sub-step has a dozen of them (0.06 ms per step:
    # 2 % of this grid's step, 10 % of its 8-GPU tile's), so they run in a diagnostic window of their own after the timed one.
    lib.icar_hip_timing_groups(d.ctx, b"advect")
    lib.icar_hip_timing_enable(d.ctx, 0 if args.no_kernel_timers else 1); lib.icar_hip_timing_reset(d.ctx)
    t0 = time.perf_counter()
    dt = run_steps(d, opt, args.steps)       # exactly K steps: K x (icar_hip_update_dt + icar_hip_substep), issued from C
    barrier()
    elapsed = time.perf_counter() - t0
    tot = ctypes.c_double(); n = ctypes.c_int()
    lib.icar_hip_timing_read(d.ctx, b"advect", ctypes.byref(tot), ctypes.byref(n))
    adv_ms = tot.value / max(n.value, 1)
    diag_steps = max(1, min(args.steps, 10))
    lib.icar_hip_timing_groups(d.ctx, b"mp,winds"); lib.icar_hip_timing_reset(d.ctx)
    run_steps(d, opt, diag_steps)            # untimed: the microphysics and advection-setup timers of the steps that follow
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    own_cells = (g.ite - g.its + 1) * (g.jte - g.jts + 1) * args.nz
    cells_t = torch.tensor([float(own_cells)], dtype=torch.float64, device=red_device)
    if world > 1:
        dist.all_reduce(cells_t)
    total_cells = float(cells_t.item())

    # roofline of the dominant kernel group: the MPDATA advection launches, HIP events on the ctx stream
    lib.icar_hip_timing_read(d.ctx, b"mp", ctypes.byref(tot), ctypes.byref(n))
    mp_ms_step = tot.value / diag_steps
    lib.icar_hip_timing_read(d.ctx, b"winds", ctypes.byref(tot), ctypes.byref(n))     # k_setup_winds + k_mpdata_coef, beside the interior mp
    winds_ms = tot.value / max(n.value, 1)
    mem_cells = d.nx * d.ny * d.nz
    alg_bytes = mem_cells * (8 * nscal + 16)            # B_adv = 8N+16 bytes per cell
    achieved = alg_bytes / (adv_ms * 1e-3) / 1e9 if adv_ms > 0 else 0.0
    # HBM bytes per advect() call from the PMC passes (profiles/run_profiles.sh: FETCH_SIZE x2 + WRITE_SIZE, separate
    # runs of this same command): counters cannot be read inside this process, so the figure is attached only when it was
    # taken on THIS configuration and THIS kernel generation; otherwise null.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "advect_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            want = {"nx": d.nx, "ny": d.ny, "nz": d.nz, "adv": args.adv, "nscalars": nscal, "kernels": ADVECT_KERNELS[args.adv],
                    "generation": KERNEL_GENERATION}
            if tj.get("config") == want and not args.mpdata_exact:      # (the recorded passes are the fused kernel's)
                traffic = tj.get("hbm_bytes_per_advect_call")
                traffic_source = "profiles/advect_traffic.json (PMC passes of an earlier run of this configuration and kernel generation)"
        except Exception:
            traffic = None

    stream_gbs = None
    if rank == 0:
        # the spec peak next to what a plain streaming copy sustains on THIS box (read + write of 1 GiB,
        # torch's device-to-device copy kernel), measured outside the timed region
        try:
            a = torch.empty(1 << 28, dtype=torch.float32, device=device); b = torch.empty_like(a)
            b.copy_(a); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                b.copy_(a)
            e1.record(); torch.cuda.synchronize()
            stream_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del a, b
        except Exception:  # pragma: no cover
            stream_gbs = None

    if rank == 0:
        # liveness: how much of the tile the microphysics is doing work in
        qc = d.get("cloud_water_mass"); qr = d.get("rain_mass")
        active = float((((qc > 1e-8) | (qr > 1e-8)).any(axis=1)).mean())
    # The cost of a step depends on the model state through the microphysics (the ideal case rains out): outside the timed region,
    # the same K steps once more after 100 further steps, as a second window next to the headline one (informational).
    later = None
    if args.steps > 0 and not args.no_later_window and not args.no_cpu_baseline:      # (the full line only, like the traffic probe)
        run_steps(d, opt, 100)
        barrier()
        t1 = time.perf_counter()
        run_steps(d, opt, args.steps)
        barrier()
        later_ms = (time.perf_counter() - t1) / args.steps * 1e3
        if world > 1:
            t = torch.tensor([later_ms], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            later_ms = float(t.item())
        if rank == 0:
            qc = d.get("cloud_water_mass"); qr = d.get("rain_mass")
            later = {"after_steps": args.warmup + args.steps + diag_steps + 100, "steps": args.steps, "ms_per_step": later_ms,
                     "mp_active_column_fraction": float((((qc > 1e-8) | (qr > 1e-8)).any(axis=1)).mean())}
    if rank == 0:
        out = {
            "metric": "grid-cell updates/sec (advection+microphysics)",
            "value": total_cells * args.steps / elapsed,
            "unit": "grid-cell updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.nx}x{args.ny}x{args.nz} global grid" if args.scaling == "strong" else
                                    f"{args.nx}x{args.ny}x{args.nz} owned cells per GPU ({g.nx_global}x{g.ny_global}x{args.nz} global)") +
                                   f", {('mpdata order-2+FCT (reference operation order, bit-exact mode)' if args.mpdata_exact else 'mpdata order-2+FCT') if args.adv == 'mpdata' else 'upwind'} advection of "
                                   f"{nscal} scalars + {args.mp} microphysics, ideal hill case (SURVEY 8d)",
                       "global_grid": [g.nx_global, g.ny_global, args.nz],
                       "tile_memory": [d.nx, d.nz, d.ny], "decomposition": f"{g.ximages}x{g.yimages}",
                       # the transport the library actually opened (icar_hip_comm_kind), not the one asked for
                       "backend": {capi.COMM_RCCL: "rccl", capi.COMM_HOST: "host-staged (functional path, not a performance number)"}.get(kind, "none") if in_group else "none",
                       **({"transport_note": d.comm.transport_note} if getattr(d.comm, "transport_note", None) else {}),
                       "halo": ("one ncclSend/ncclRecv group per step issued by the library (icar_hip_halo_send)" if kind == capi.COMM_RCCL else
                                "one message per neighbour through pinned host memory (icar_hip_comm_init_host)") + ", strips+pack on the second stream beside the interior mp" if world > 1
                               else "periodic self-exchange (pack + unpack of 4 edges, no transport), strips+pack on the second stream beside the interior mp",
                       "ranks_seen": ranks_seen, "halo_check": halo_check,
                       # what the timed path computes: the microphysics, diagnostics, forcing and halos bit-identical to the CPU reference;
                       # MPDATA either the fused kernel (every cell within 1e-5 of the local field scale per step, measured <= 8.1e-7: profiles/r06_parity.json) or,
                       # with --mpdata-exact, the reference's operation order (bit-identical; tests/test_gpu_trajectory.py)
                       "mpdata_arithmetic": ("reference operation order (bit-identical)" if args.mpdata_exact else "fused kernel (<= 1e-5 of the local scale per step)") if args.adv == "mpdata" else "n/a",
                       "dt_s": dt, "mp_active_column_fraction": active},
            "later_window": later,
            "roofline": {"bound": "hbm", "kernel": "advect (k_upwind_pass + k_mpx_velocities + k_mpx_limit_donor: icar_hip_mpdata_exact)" if (args.mpdata_exact and args.adv == "mpdata")
                                                   else f"advect ({ADVECT_KERNELS[args.adv]})",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": alg_bytes, "avg_ms": adv_ms,
                         # mp_ms_per_step / setup_ms_per_step: over the `timer_window_steps` steps that FOLLOW the timed region (see above)
                         "mp_ms_per_step": mp_ms_step, "timer_window_steps": diag_steps,
                         # the once-per-step setup of the advection (Courant winds + the scalar-independent MPDATA coefficients),
                         # issued beside the interior microphysics; not part of avg_ms
                         "setup_ms_per_step": winds_ms,
                         # the advection GROUP of a step = advect() + its setup kernels, whatever they run beside: same algorithmic
                         # bytes (the setup reads nothing the 8N+16 model does not already count) over the sum of the two timers
                         "group_ms": adv_ms + winds_ms,
                         "frac_group": (alg_bytes / ((adv_ms + winds_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if adv_ms + winds_ms > 0 else None,
                         # informational: scalar-cell updates/s of the advection alone, and the measured
                         # streaming-copy bandwidth of this box beside the spec peak that `frac` uses
                         "advect_scalar_cell_updates_per_s": (mem_cells * nscal / (adv_ms * 1e-3)) if adv_ms > 0 else None,
                         "measured_copy_GBps": stream_gbs,
                         "frac_of_measured_copy": (achieved / stream_gbs) if stream_gbs else None},
            # informational: the microphysics is VALU-bound, its share of the step and its algorithmic traffic
            "microphysics": {"kernel": {"thompson": "k_thompson_pack", "simple": "k_mp_simple_pack", "wsm3": "k_wsm3", "wsm6": "k_w6"}.get(args.mp, "none"), "bound": "valu",
                             "ms_per_step": mp_ms_step,
                             "algorithmic_GBps": (mem_cells * (84 if args.mp == "thompson" else 56) / (mp_ms_step * 1e-3) / 1e9) if mp_ms_step > 0 else 0.0},
        }
        if world == 1 and not args.no_traffic_probe and not args.no_cpu_baseline and not args.mpdata_exact:        # (the full line only: profiling scripts pass --no-cpu-baseline)
            # roofline.traffic measured NOW on this box (counters cannot be read inside this process: two child runs under rocprofv3)
            t = probe_traffic(args)
            if t is not None:
                out["roofline"].update({"traffic": t[0], "traffic_read_bytes": t[1], "traffic_write_bytes": t[2],
                                        "traffic_setup_bytes": t[3], "traffic_group": t[0] + t[3],
                                        "traffic_group_over_algorithmic": (t[0] + t[3]) / alg_bytes,
                                        "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x 2 for gfx950) over "
                                                          "3-step child runs of this configuration, mean per dispatch of " + ADVECT_KERNELS[args.adv]})
        if not args.no_cpu_baseline and world == 1:          # the CPU legs are timed at N=1 only (the other ranks would idle at the barrier)
            out["cpu_baseline"] = cpu_baseline(args, nscal)
            r = cpu_reference(args, nscal)
            if r is not None:
                out["cpu_reference"] = r
        print(json.dumps(out), flush=True)
    d.close()
    if in_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
