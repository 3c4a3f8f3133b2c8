"""Build libicar_hip.so (hipcc, gfx950 only) in-tree: icar_amd/lib/libicar_hip.so.

`python -m icar_amd.build` or icar_amd.build.build().  Cross-compiles without a GPU.
-ffp-contract=off + IEEE divide/sqrt keep the upwind scheme, the microphysics and the streaming rows
bit-identical to the CPU oracle; mpdata.hip alone allows fma contraction and 1-ulp reciprocals and is held to the 1e-5
tolerance on every cell (tests/test_gpu_advect.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libicar_hip.so")
SOURCES = ["capi.hip", "advect.hip", "mpdata.hip", "mpdata_exact.hip", "mp_simple.hip", "mp_thompson.hip", "thompson_tables.hip", "step.hip", "linear_winds.hip", "iterative_winds.hip", "mp_wsm3.hip", "mp_wsm6.hip", "comm.hip", "timestep.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# finite data only: drop the NaN-canonicalisation v_max x,x,x in front of every fmin/fmax (no effect on finite results)
PER_FILE_FLAGS = {"advect.hip": ["-fno-honor-nans"],
                  # the fused MPDATA kernel is held to the 1e-5 tolerance, not to bit equality: fma contraction allowed
                  # ... and no SLP packing: the vectoriser's v_pk_add / mul / fma_f32 pairs are assembled with more v_mov than they
                  # save (profiles/r05_steps.md)
                  "mpdata.hip": ["-fno-honor-nans", "-ffp-contract=fast", "-fno-slp-vectorize"],
                  # k_thompson_pack runs at 4 waves per SIMD with ~50 VGPRs in scratch.  The one miscompile seen in this code (round 4,
                  # a since-removed kernel) needed SGPRs spilled into VGPR lanes on top of VGPR spills: SGPR spills, should a compiler
                  # ever produce them here, go to memory instead
                  # ... and the kernel waits on its dependent chains more than on the VALU (round 5: -8 % VALU instructions change nothing):
                  # the scheduler that orders for instruction-level parallelism instead of register pressure (36 instead of 38
                  # VGPRs in scratch as it happens) is 1-2 % faster alone, 0.5 % per step; same bits (no contraction, IEEE order)
                  "mp_thompson.hip": ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=false", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",     # (a later -ffp-contract wins)
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "icar_hip.h"))
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.stderr.strip():
                sys.stderr.write(r.stderr)                # warnings of every file, and the diagnostics of a failed compile
            if r.returncode != 0:
                if os.path.exists(obj):
                    os.remove(obj)                        # (a later build() must not take the object for up to date)
                raise RuntimeError(f"{s}: hipcc failed with status {r.returncode} (diagnostics above)")
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def kernel_resources(src, extra=()):
    """{kernel name: {"VGPRs": .., "ScratchSize [bytes/lane]": .., "Occupancy [waves/SIMD]": .., "SGPRs Spill": .., "VGPRs Spill": ..,
    "TotalSGPRs": .., "AGPRs": .., "LDS Size [bytes/block]": ..}} of one source of SOURCES, compiled with the product's flags: the
    compiler's -Rpass-analysis=kernel-resource-usage remarks.  What the performance of k_mpdata_fused and k_thompson_pack rests on
    (two / four waves per SIMD, nothing of the march in scratch) is decided by the register allocator, not by the source:
    tests/test_kernel_resources.py holds the numbers, profiles/micro/resources.py prints them."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory(prefix="icar_res_") as tmp:
        cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(src, []) + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "o.o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{src}: hipcc failed:\n{r.stderr[-2000:]}")
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
            cur = out.setdefault(name, {})
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            try:
                cur[k.strip()] = int(v.strip())
            except ValueError:
                cur[k.strip()] = v.strip()
    return out


FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
FDIR = os.path.join(HERE, "fortran")
DEMO = os.path.join(LIBDIR, "icar_hip_demo")
STEP_DEMO = os.path.join(LIBDIR, "icar_hip_step_demo")
TILES_DEMO = os.path.join(LIBDIR, "icar_hip_tiles_demo")


def build_fortran_host(force=False, verbose=False):
    """Fortran 2008 host side (iso_c_binding module + demo driver); needs flang (present in the image)."""
    if not os.path.exists(FLANG):
        return None
    mod = os.path.join(FDIR, "icar_hip_mod.f90"); demo = os.path.join(FDIR, "icar_hip_demo.f90")
    sdemo = os.path.join(FDIR, "icar_hip_step_demo.f90"); tdemo = os.path.join(FDIR, "icar_hip_tiles_demo.f90")
    if not (force or _stale(DEMO, [mod, demo, LIB]) or _stale(STEP_DEMO, [mod, sdemo, LIB]) or _stale(TILES_DEMO, [mod, tdemo, LIB])):
        return DEMO
    obj = os.path.join(LIBDIR, "icar_hip_mod.o")
    cmds = [[FLANG, "-O2", "-c", mod, "-o", obj, "-module-dir", LIBDIR],
            [FLANG, "-O2", "-I" + LIBDIR, demo, obj, "-L" + LIBDIR, "-licar_hip", "-Wl,-rpath,$ORIGIN", "-o", DEMO],
            [FLANG, "-O2", "-I" + LIBDIR, sdemo, obj, "-L" + LIBDIR, "-licar_hip", "-Wl,-rpath,$ORIGIN", "-o", STEP_DEMO],
            [FLANG, "-O2", "-I" + LIBDIR, tdemo, obj, "-L" + LIBDIR, "-licar_hip", "-Wl,-rpath,$ORIGIN", "-o", TILES_DEMO]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return DEMO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_fortran_host(force="--force" in sys.argv, verbose=True))
