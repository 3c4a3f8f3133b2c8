"""Times the UNMODIFIED reference kernels (oracle/_ref: adv_mpdata.f90 + mp_thompson.f90 compiled by oracle/build_ref.sh with
flang -O2, no OpenMP -> one core) on one step of bench.py's workload, IN THE BUILD CONTAINER -- the one place where
oracle/build_ref.sh can rebuild that library from /root/reference -- and records the result in profiles/cpu_reference.json.
bench.py attaches the record to its line as `cpu_reference` ("kind": "reference (recorded)"); nothing of the reference is
executed on the GPU box.  Run: python profiles/measure_cpu_reference.py [--nx 512 --ny 512 --nz 40]
Test infrastructure (it imports oracle/), not product."""
import argparse, json, os, platform, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np
    from oracle import ref
    from icar_amd import ideal
    nx, ny, nz = a.nx, a.ny, a.nz
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.4)).astype(np.float32)
    dt = min(ideal.cfl_dt(c), 120.0)
    names = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature", "cloud_ice", "graupel", "ice_number", "rain_number"]
    q = np.stack([c[n] for n in names]).copy()
    t0 = time.perf_counter()
    ref.thompson_init(workdir=a.child)                 # the reference's own table build (cached as .dat in workdir)
    t_init = time.perf_counter() - t0
    acc = [np.zeros((ny, nx), np.float32) for _ in range(5)]
    t0 = time.perf_counter()
    ref.thompson(c["water_vapor"], c["cloud_water"], c["rain"], c["cloud_ice"], c["snow"], c["graupel"], c["ice_number"],
                 c["rain_number"], c["potential_temperature"], c["exner"], c["pressure"], c["dz_mass"], dt, *acc,
                 1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
    t_mp = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
               c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
    t_adv = time.perf_counter() - t0
    print("\n" + json.dumps({"t_init": t_init, "t_mp": t_mp, "t_adv": t_adv}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=512); ap.add_argument("--ny", type=int, default=512); ap.add_argument("--nz", type=int, default=40)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        return child(a)
    import resource
    def unlimited():        # adv_mpdata.f90:365-368 keeps four full-grid temporaries on the stack: `ulimit -s unlimited` like any ICAR run
        hard = resource.getrlimit(resource.RLIMIT_STACK)[1]; resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))
    tmp = tempfile.mkdtemp(prefix="icar_ref_")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tmp, "--nx", str(a.nx), "--ny", str(a.ny), "--nz", str(a.nz)],
                       capture_output=True, text=True, timeout=3600, preexec_fn=unlimited)
    t = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    el = t["t_mp"] + t["t_adv"]
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    rec = {"value": (a.nx - 2) * (a.ny - 2) * a.nz / el, "unit": "grid-cell updates/s", "cores": 1, "kind": "reference (recorded)",
           "where": f"build container, {cpu[0] if cpu else platform.processor()} ({len(cpu)} logical CPUs visible, one used)",
           "sample": f"1 step of {a.nx}x{a.ny}x{a.nz}: the reference's own mp_thompson.f90 ({t['t_mp']:.1f} s) + adv_mpdata.f90 order 2 + FCT on 9 scalars "
                     f"({t['t_adv']:.1f} s), compiled unmodified by oracle/build_ref.sh (flang -O2, no OpenMP); table build {t['t_init']:.0f} s not counted",
           "config": {"nx": a.nx, "ny": a.ny, "nz": a.nz, "adv": "mpdata", "mp": "thompson", "nscalars": 9},
           "recorded": time.strftime("%Y-%m-%d")}
    json.dump(rec, open(os.path.join(ROOT, "profiles", "cpu_reference.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
