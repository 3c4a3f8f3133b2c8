"""Wall-clock shares of the sections of the packed Thompson kernel (measurement aid, not product).

python profiles/micro/th_sections.py            -> icar_amd/lib/ab/lib_thprof.so
A COPY of mp_thompson.hip / thompson_lane.inc gets `TH_STAMP(n)` calls in front of anchor lines: s_memtime of the wave, the
difference to the wave's previous stamp added to an LDS slot by lane 0, the block's slots flushed to a global array at the end;
the host prints the totals at exit.  The product sources are not touched.  Run any driver with
ICAR_HIP_LIB=icar_amd/lib/ab/lib_thprof.so; the table goes to stderr.
"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
from icar_amd import build as B

# (name of the section that ENDS at the anchor, anchor substring, occurrence (1-based), file)
ANCHORS = [
    ("launch: tile index, exchange areas cleared, lds tables", "    const int j = tall ? (x.active ?", 1, "hip"),
    ("index + loads issued", "th_column_lane(T, x, nk, dt, dz1d", 2, "hip"),
    ("-> start of level code", "dtsave = dt; odt = 1.f / dt; odts", 1, "inc"),
    ("working variables (first use of the loads; lami, lamr)", "/* no_micro needs the ice saturation", 1, "inc"),
    ("ice saturation", "    /* :1363 (nothing to do in the whole column group", 1, "inc"),
    ("chain 0: N0_exp, any + suffix-min exchange", "        GRAUPEL_SLOPE(vmin_)", 1, "inc"),
    ("chain 0: slope (2 pow)", "/* ---- per-level phase A", 1, "inc"),
    ("A: saturation, diffu, visco", "            if (L_qs) {", 1, "inc"),
    ("A: snow moments", "/* rain slope, mean volume diameter, intercept", 1, "inc"),
    ("A: rain slope, N0_r, warm rain", "        vts_boost = 1.5f;", 1, "inc"),
    ("A: table indices", "/* deposition/sublimation prefactor", 1, "inc"),
    ("A: deposition prefactor", "/* snow / graupel collecting cloud water", 1, "inc"),
    ("A: snow / graupel collecting cloud water", "/* rain collecting snow / graupel", 1, "inc"),
    ("A: rain collecting snow / graupel (tables)", "if (temp < T_0) {      /* :1789-1949", 1, "inc"),
    ("A: sub-zero processes | melting", "            sump = (float)(pri_inu + pri_ide", 1, "inc"),
    ("A: conservation", "            orho = 1.f / rho;", 1, "inc"),
    ("A: tendencies, ice / rain number checks", "/* ---- per-level phase B", 1, "inc"),
    ("B: TAU+1 thermodynamics", "if ((qc1d + qcten * dt) > R1)", 1, "inc"),
    ("B: TAU+1 contents, rain slope", "            if (L_qs) {", 4, "inc"),
    ("B: snow moments", "/* input of the second graupel chain", 1, "inc"),
    ("B: xslw, N0_r", "if ((ssatw > eps) || (ssatw < -eps && L_qc))", 1, "inc"),
    ("B: condensation (Newton)", "if ((ssatw < -eps) && L_qr && (!(prw_vcd > 0.)))", 1, "inc"),
    ("B: rain evaporation", "    /* ---- :2379-2391 second graupel chain and", 1, "inc"),
    ("chain 1 N0_exp + fall speeds of rain, ice, snow", "            const float pv_[5]", 1, "inc"),
    ("exchange 1 (post, barrier)", "        const int k_ = x.k;", 1, "inc"),
    ("gathers: rain, ice, snow", "        GRAUPEL_SLOPE(x.min_fall1())", 1, "inc"),
    ("chain 1 minimum, slope, graupel speed", "        x.post_fall2(a_g);", 1, "inc"),
    ("exchange 2 (post, barrier)", "            const int kg = x.above(3, k_)", 1, "inc"),
    ("gather graupel, sub-step counts, hand-off", "x.plan4(h.c4", 1, "inc"),
    ("sedimentation plan exchange", "odzq = 1.f / dzq; orho = 1.f / rho;", 1, "inc"),
    ("sedimentation loop", "    h.rr = rr; h.nr = nr; h.ri = ri; h.ni = ni; h.rs = rs; h.rg = rg;\n    h.qrten = qrten; h.nrten = nrten; h.qiten = qiten; h.niten = niten; h.qsten = qsten; h.qgten = qgten;\n}", 1, "inc"),
    ("-> finish", "th_level_finish(T, dt, h, qv1d", 1, "inc"),
    ("melt / freeze, apply", "    if (!x.active) return;\n    if (x.k == 0) {", 1, "hip"),
]

STAMP = r'''
__device__ unsigned long long th_prof_acc[64];
__device__ __forceinline__ void th_stamp(int n)
{
    __shared__ unsigned long long th_last[16], th_acc[64];
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const int w = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) {
        if (n < 0) { if (threadIdx.x == 0) for (int i = 0; i < 64; ++i) th_acc[i] = 0ull; th_last[w] = t; }
        else if (n < 64) { atomicAdd(&th_acc[n], t - th_last[w]); th_last[w] = t; }
    }
    if (n == 64) { __syncthreads(); if (threadIdx.x < 64 && th_acc[threadIdx.x]) atomicAdd(&th_prof_acc[threadIdx.x], th_acc[threadIdx.x]); }
}
#define TH_STAMP(n) th_stamp(n);
'''

DUMP = r'''
        if (getenv("ICAR_TH_PROF")) {
            static unsigned long long tot[64]; static bool reg = false;
            unsigned long long cur[64], zero[64] = {0};
            hipDeviceSynchronize();
            hipMemcpyFromSymbol(cur, HIP_SYMBOL(th_prof_acc), sizeof(cur));
            hipMemcpyToSymbol(HIP_SYMBOL(th_prof_acc), zero, sizeof(zero));
            for (int i = 0; i < 64; ++i) tot[i] += cur[i];
            if (!reg) { reg = true; atexit([] { unsigned long long s = 0; for (int i = 0; i < 64; ++i) s += tot[i];
                for (int i = 0; i < 64; ++i) if (tot[i]) fprintf(stderr, "TH_PROF %2d %14llu %6.2f%%\n", i, tot[i], 100.0 * tot[i] / s); }); }
        }
'''


def insert(text, anchor, occ, payload):
    pos = -1
    for _ in range(occ):
        pos = text.find(anchor, pos + 1)
        if pos < 0:
            raise SystemExit(f"anchor not found: {anchor!r} #{occ}")
    if "\n" not in anchor:                      # go to the start of the line
        pos = text.rfind("\n", 0, pos) + 1
    return text[:pos] + payload + text[pos:]


def main():
    out = "/tmp/thprof"
    os.makedirs(out, exist_ok=True)
    src = {"hip": open(os.path.join(B.CSRC, "mp_thompson.hip")).read(), "inc": open(os.path.join(B.CSRC, "thompson_lane.inc")).read()}
    # the anchors are resolved on the ORIGINAL text one by one; inserting back to front keeps earlier positions valid
    todo = {"hip": [], "inc": []}
    for n, (name, a, occ, f) in enumerate(ANCHORS):
        t = src[f]; pos = -1
        for _ in range(occ):
            pos = t.find(a, pos + 1)
            if pos < 0:
                raise SystemExit(f"anchor not found: {a!r} #{occ}")
        if "\n" not in a:
            pos = t.rfind("\n", 0, pos) + 1
        todo[f].append((pos, n))
    for f in todo:
        t = src[f]
        for pos, n in sorted(todo[f], reverse=True):
            t = t[:pos] + f"TH_STAMP({n})\n" + t[pos:]
        src[f] = t
    hip = src["hip"]
    hip = hip.replace('#include "thompson_lane.inc"', STAMP + '#include "thompson_lane_prof.inc"', 1)
    # first stamp of a block: reset, before the table copy; last: flush (before the early return of idle threads)
    hip = insert(hip, "    // several (its..ite, jts..jte) tiles in one launch", 1, "    TH_STAMP(-1) __syncthreads();\n")
    hip = insert(hip, "    if (!x.active) return;\n    if (x.k == 0) {", 1, "    TH_STAMP(64)\n")
    hip = insert(hip, "        HIPCHK(hipGetLastError());\n        return 0;\n    }\n    if (nk > 64)", 1, DUMP)
    hip = hip.replace("#include <cstring>", "#include <cstring>\n#include <cstdio>", 1)
    open(os.path.join(out, "mp_thompson_prof.hip"), "w").write(hip)
    open(os.path.join(out, "thompson_lane_prof.inc"), "w").write(src["inc"])
    with open(os.path.join(out, "sections.txt"), "w") as fh:
        for n, (name, *_r) in enumerate(ANCHORS):
            fh.write(f"{n:2d} {name}\n")
    subprocess.check_call([sys.executable, "profiles/micro/build_ab.py", "mp_thompson.hip", f"thprof@{out}/mp_thompson_prof.hip:-I{out}"])
    print(open(os.path.join(out, "sections.txt")).read())


if __name__ == "__main__":
    main()
