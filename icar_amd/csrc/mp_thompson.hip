// icar_amd/csrc/mp_thompson.hip -- Thompson et al. (2008) bulk microphysics on gfx950 (rows M2/M3).
//
// Reference: src/physics/mp_thompson.f90 -- mp_gt_driver :772-1044 (column gather/scatter, i_end/j_end
// clipping, precipitation accumulation, the qv floor of :997-1010) and mp_thompson :1057-2844 (the
// 1-D column physics).  State is REAL(4), rates and lookup-table values REAL(8) exactly as in the reference; table indices are
// integer-exact; REAL(4) and DOUBLE PRECISION transcendentals restate the glibc 2.35 builds the reference links (glibc_flt32.h,
// glibc_dbl64.h): every column is bit-identical to the compiled reference's (tests/test_gpu_thompson.py).
//
// Layout of the product kernel k_thompson_pack: ONE LEVEL PER THREAD -- a block packs several whole columns, a thread owns one
// (column, level) cell and runs the level code (thompson_lane.inc) on registers; what crosses levels (the graupel N0 chain, the
// fall-speed carry-down, the four sedimentation sweeps, the column's "anything to do" flags) goes through five one-barrier LDS
// exchanges per column (column_comm.h).  The ~130 per-level work arrays of the reference are registers of the level's thread.
// (k_thompson_lane is the same level code with one column per WAVE, one level per lane: the fallback for level counts the packing
// does not cover.)
// This departs from north_star's "one column per lane, lanes along i".  That layout -- a lane marches through its column with the
// cross-level state in lane-interleaved private arrays -- was built and measured in round 4: 2.40 ms against 1.77 ms for the
// level-per-thread form at 512 x 512 x 40 (profiles/r04_thompson_layout.md), and then removed.  With 40 levels a column per lane
// leaves 512 x 512 = 262144 lanes, 4 waves per SIMD of a kernel that wants > 128 VGPRs per lane; a level per thread gives 40 x as
// many lanes to hide the FP64 table and division latencies with.  Loads and stores of a wave are rows of consecutive i at one level
// in both layouts (memory order (i, k, j)).
#include "ctx.h"
#include "thompson_math.h"     // the level code's transcendentals and decade indices (shared with tests/support/th_probe.hip)
#include "column_comm.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstddef>

const ThState *icar_thompson_device_state(icar_hip_ctx *c);
const ThState *icar_thompson_host_state(icar_hip_ctx *c);


namespace {

/* x**3.0 with a PARAMETER exponent is expanded to multiplications by flang (verified: the tables are
 * bit-identical to the reference only with x*x*x) */
__device__ __forceinline__ float cube_f(float x) { return x * x * x; }

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ float rslf(float P, float T)
{   /* :3776-3805 */
    const float C0 = .611583699E03f, C1 = .444606896E02f, C2 = .143177157E01f, C3 = .264224321E-1f, C4 = .299291081E-3f,
                C5 = .203154182E-5f, C6 = .702620698E-8f, C7 = .379534310E-11f, C8 = -.321582393E-13f;
    const float X = fmaxf(-80.f, T - 273.16f);
    const float ESL = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    return .622f * ESL / (P - ESL);
}

__device__ __forceinline__ float rsif(float P, float T)
{   /* :3810-3835 */
    const float C0 = .609868993E03f, C1 = .499320233E02f, C2 = .184672631E01f, C3 = .402737184E-1f, C4 = .565392987E-3f,
                C5 = .521693933E-5f, C6 = .307839583E-7f, C7 = .105785160E-9f, C8 = .161444444E-12f;
    const float X = fmaxf(-80.f, T - 273.16f);
    const float ESI = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    return .622f * ESI / (P - ESI);
}

/* Field et al. (2005) moment polynomial in REAL arithmetic (:1379-1449); b = moment order */
__device__ __forceinline__ float snow_poly_f(const float *s, float tc0, float b)
{
    return s[0] + s[1] * tc0 + s[2] * b + s[3] * tc0 * b + s[4] * tc0 * tc0 + s[5] * b * b + s[6] * tc0 * tc0 * b
         + s[7] * tc0 * b * b + s[8] * tc0 * tc0 * tc0 + s[9] * b * b * b;
}


#define T4S(tab) (T->tab[(idx_s - 1) + NTB_S * ((idx_t - 1) + NTB_T * ((size_t)(idx_r1 - 1) + NTB_R1 * (idx_r - 1)))])
#define T4G(tab) (T->tab[(idx_g1 - 1) + NTB_G1 * ((idx_g - 1) + NTB_G * ((size_t)(idx_r1 - 1) + NTB_R1 * (idx_r - 1)))])
#define T3R(tab) (T->tab[(idx_r - 1) + NTB_R * ((idx_r1 - 1) + NTB_R1 * (size_t)(idx_tc - 1))])
#define T2C(tab) (T->tab[(idx_c - 1) + NTB_C * (size_t)(idx_tc - 1)])
#define T2I(tab) (T->tab[(idx_i - 1) + NTB_I * (size_t)(idx_i1 - 1)])


#include "thompson_lane.inc"

// one column per wave (4 columns per 256-thread block), one level per lane; see thompson_lane.inc
__global__ void __launch_bounds__(256, 4)   // 4 waves/SIMD (128 VGPRs, 252 B spill) measured best of 2..8: 7.2/6.0/5.6/6.2/6.3/9.2 ms
k_thompson_lane(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                float dt, int i0, int i1, int j0, int k0, int nk)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const int lane = threadIdx.x & 63;
    const int i = i0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = j0 + blockIdx.y;
    if (i > i1) return;                                   // wave-uniform
    const int kk = lane < nk ? lane : nk - 1;
    const int c = d.idx(i, k0 + kk, j);
    const float pi_ = pii[c];
    float t1d = th[c] * pi_, p1d = p[c], dz1d = dz[c], qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c],
          qg1d = qg[c], ni1d = ni[c], nr1d = nr[c];
    float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
    WaveComm x(lane, nk);
    th_column_lane(T, x, nk, dt, dz1d, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, pptrain, pptsnow, pptgraul, pptice);
    if (lane == 0) {
        const int c2 = i + d.nx * j;
        const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
        const float snownc = 0.f + pptsnow + pptice;
        const float graupelnc = 0.f + pptgraul;
        rain_acc[c2] = rain_acc[c2] + rainnc;
        snow_acc[c2] = snow_acc[c2] + snownc;
        graupel_acc[c2] = graupel_acc[c2] + graupelnc;
    }
    if (lane < nk) {
        qv[c] = (qv1d < 1.E-7f) ? 1.E-7f : qv1d;          // :997-1010
        qc[c] = qc1d; qi[c] = qi1d; qr[c] = qr1d; qs[c] = qs1d; qg[c] = qg1d; ni[c] = ni1d; nr[c] = nr1d;
        th[c] = t1d / pi_;
    }
}

struct ThTiles { int n, i0[4], i1[4], j0[4], j1[4], tall[4], ib0[4], nbx[4], off[5], xcd_run; };
// The leading arguments of k_thompson_pack as they lie in its kernel-argument segment (by-value parameters in order, each at its
// natural alignment = this struct's layout).  The kernel reads the field pointers a second time from that segment behind the level
// code, for the stores, instead of keeping them (or nine 64-bit store addresses) in registers through the level code.  The
// parameters themselves stay separate `__restrict__` pointers: as members of one struct argument they lose `noalias`, and the ~180
// scalar loads of scheme parameters through T become vector loads (measured).
struct ThPackArgs {
    Dims d; const ThState *T;
    float *qv, *qc, *qr, *qi, *qs, *qg, *ni, *nr, *th; const float *pii, *p, *dz;
    double *rain_acc, *snow_acc, *graupel_acc;
};
static_assert(offsetof(ThPackArgs, T) == ((sizeof(Dims) + 7) & ~(size_t)7) && offsetof(ThPackArgs, qv) == offsetof(ThPackArgs, T) + 8
              && offsetof(ThPackArgs, graupel_acc) == offsetof(ThPackArgs, qv) + 14 * 8, "ThPackArgs mirrors the kernel's argument list");

// cpb whole columns per block (aligned to multiples of cpb in i), thread = level*cpb + column (thompson_lane.inc: BlockComm)
// MAXT = largest block this instantiation is launched with.  Blocks of up to 512 threads (columns of up to 512 levels) get the
// register budget of 3 waves per SIMD (168 VGPRs, no spills); at 128 VGPRs / 4 waves the kernel ran exactly as fast but spilled
// 62 VGPRs -- 4 GB of scratch traffic per launch against 0.9 GB of algorithmic bytes.  1024-thread blocks need the 128.
template <int MAXT>
// Waves per SIMD of the 256-thread launch: until the DOUBLE PRECISION functions became glibc's (table look-ups from LDS: more
// latency to hide, and no scalar registers held for polynomial coefficients any more) three waves at 168 VGPRs beat four at 128
// (1.93 vs 1.98 ms); since then four win -- 1.77 against 1.86 ms alone, 3.10 against 3.15 ms per step at 512 x 512 x 40 (53 VGPRs
// in scratch, no SGPR spills), equal on the small tiles (profiles/r04_steps.md).
__global__ void __launch_bounds__(MAXT, 4)
k_thompson_pack(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                float dt, ThTiles tl, int k0, int nk, int cpb)
{
    extern __shared__ double lds_pack[];
    // several (its..ite, jts..jte) tiles in one launch (process_halo's four strips): block -> tile by prefix offsets
    // XCD-aware order: workgroups go to the 8 XCDs round-robin; neighbouring column groups share 64-B lines (a group is
    // 24 B wide at nz = 40), so each XCD takes runs of XCD_RUN consecutive groups and the shared lines hit in its L2.
    int bid = (int)blockIdx.x;
    {
        const int run = tl.xcd_run, super = run * 8, sc = bid / super, w = bid % super;
        if ((sc + 1) * super <= (int)gridDim.x) bid = sc * super + (w % 8) * run + w / 8;
    }
    int t = 0;
    while (t + 1 < tl.n && bid >= tl.off[t + 1]) ++t;
    const int local = bid - tl.off[t];
    const int i0 = tl.i0[t], i1 = tl.i1[t];
    // a block's cpb column slots run along i (aligned to multiples of cpb), except in a tile ONE column wide (the west / east
    // strips of process_halo), where they run along j: one busy column of six per block otherwise
    const bool tall = tl.tall[t] != 0;
    const int first = tall ? tl.j0[t] + local * cpb : (tl.ib0[t] + local % tl.nbx[t]) * cpb;
    BlockComm x(lds_pack, threadIdx.x, blockDim.x, cpb, nk, tall ? 0 : i0 - first, tall ? tl.j1[t] - first : i1 - first);
    x.th_init((double)TH_gonv_max);                       // the exchange areas that are combined with min / atomics
    th_lds_init(threadIdx.x, blockDim.x);                 // (ends with the block barrier that also publishes th_init's stores)
    const int j = tall ? (x.active ? first + x.col : first) : tl.j0[t] + local / tl.nbx[t];
    const int i = tall ? i0 : (x.active ? first + x.col : max(i0, min(i1, first)));
    const int c = d.idx(i, k0 + x.k, j);
    // One 32-bit byte offset for all twelve fields (base pointers stay in SGPRs: global_load v, v_off, s[base]); with `field[c]` the
    // compiler keeps a 64-bit address pair per field alive from the loads to the stores -- 24 VGPRs through the whole level code,
    // which is what it then spills (round 5).  (A field is < 4 GiB: icar_thompson_run_tiles checks.)
    const unsigned boff = (unsigned)c * 4u;
#define TH_LD(p) (*(const float *)((const char *)(p) + boff))
#define TH_ST(p, v) (*(float *)((char *)(p) + boff) = (v))
    const float pi_ = TH_LD(pii);
    float t1d = TH_LD(th) * pi_, p1d = TH_LD(p), dz1d = TH_LD(dz), qv1d = TH_LD(qv), qc1d = TH_LD(qc), qi1d = TH_LD(qi), qr1d = TH_LD(qr),
          qs1d = TH_LD(qs), qg1d = TH_LD(qg), ni1d = TH_LD(ni), nr1d = TH_LD(nr);
    float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
    th_column_lane(T, x, nk, dt, dz1d, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, pptrain, pptsnow, pptgraul, pptice);
    if (!x.active) return;
    // the argument segment again, through a pointer the compiler cannot connect with the reads at the top
    const ThPackArgs *ka = (const ThPackArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    if (x.k == 0) {
        const int c2 = i + ka->d.nx * j;
        const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
        const float snownc = 0.f + pptsnow + pptice;
        const float graupelnc = 0.f + pptgraul;
        double *ra = ka->rain_acc, *sa = ka->snow_acc, *ga = ka->graupel_acc;
        ra[c2] = ra[c2] + rainnc;
        sa[c2] = sa[c2] + snownc;
        ga[c2] = ga[c2] + graupelnc;
    }
    unsigned boff2 = boff;
    asm volatile("" : "+v"(boff2));
#undef TH_ST
#define TH_ST(p, v) (*(float *)((char *)(ka->p) + boff2) = (v))
    TH_ST(qv, (qv1d < 1.E-7f) ? 1.E-7f : qv1d);              // :997-1010
    TH_ST(qc, qc1d); TH_ST(qi, qi1d); TH_ST(qr, qr1d); TH_ST(qs, qs1d); TH_ST(qg, qg1d); TH_ST(ni, ni1d); TH_ST(nr, nr1d);
    TH_ST(th, t1d / pi_);
#undef TH_LD
#undef TH_ST
}

__global__ void k_thompson_constants(ThState *T, float rg, float xslw1)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const DK K_ = d_consts();

    T->N0_exp_default = th_graupel_N0_exp(rg, xslw1);
    T->pw_cgg_obmg = d_powf(T->cgg[2] * T->ogg2 * T->ogg1, T->obmg);
    T->pw_ccg_obmr = d_powf(T->ccg[2] * T->ocg2, T->obmr);
    T->log_Dr_span = gd_log(T->Dr[NBINS - 1] / T->Dr[0]);
    T->log_Ds_span = gd_log(T->Ds[NBINS - 1] / T->Ds[0]);
    for (int n = 0; n < TH_P10_N; ++n) T->p10[n] = powi10f(n - TH_P10_OFF);
}
}  // namespace

// called by icar_thompson_init_run once the device state exists
int icar_thompson_prepare_constants(icar_hip_ctx *c)
{
    ThState *T = const_cast<ThState *>(icar_thompson_device_state(c));
    if (!T) { icar_set_error("thompson: no device state"); return 1; }
    hipLaunchKernelGGL(k_thompson_constants, dim3(1), dim3(1), 0, c->stream, T, TH_R1, 0.01f);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_thompson_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int (*tiles)[4], int kts, int kte,
                            int ids, int ide, int jds, int jde, int kds, int kde)
{
    (void)ids; (void)jds; (void)kds; (void)kde;
    const ThState *T = icar_thompson_device_state(c);
    if (!T) { icar_set_error("thompson: call icar_hip_thompson_init first"); return 1; }
    if (ntiles < 1 || ntiles > 4) { icar_set_error("thompson: 1..4 tiles per call"); return 1; }
    if (kts < c->kms || kte > c->kme || kte < kts) { icar_set_error("thompson: levels outside memory bounds"); return 1; }
    if (c->n3 * sizeof(float) >= ((size_t)1 << 31)) { icar_set_error("thompson: a field of 2 GiB or more is not supported (32-bit byte offsets)"); return 1; }
    float *qv = icar_field_f(c, ICAR_F_WATER_VAPOR), *qc = icar_field_f(c, ICAR_F_CLOUD_WATER), *qr = icar_field_f(c, ICAR_F_RAIN);
    float *qi = icar_field_f(c, ICAR_F_CLOUD_ICE), *qs = icar_field_f(c, ICAR_F_SNOW), *qg = icar_field_f(c, ICAR_F_GRAUPEL);
    float *ni = icar_field_f(c, ICAR_F_ICE_NUMBER), *nr = icar_field_f(c, ICAR_F_RAIN_NUMBER);
    float *th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE), *pii = icar_field_f(c, ICAR_F_EXNER);
    float *p = icar_field_f(c, ICAR_F_PRESSURE), *dz = icar_field_f(c, ICAR_F_DZ_MASS);
    double *pa = (double *)icar_field_f(c, ICAR_F_PRECIPITATION, false), *sa = (double *)icar_field_f(c, ICAR_F_SNOWFALL, false);
    double *ga = (double *)icar_field_f(c, ICAR_F_GRAUPEL_ACC, false);
    if (!qv || !qc || !qr || !qi || !qs || !qg || !ni || !nr || !th || !pii || !p || !dz || !pa || !sa || !ga) return 1;
    const int nk = kte - kts + 1;
    // clip every tile like mp_gt_driver does (:821-822) and drop the empty ones
    int T4[4][4], nt_ = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int its = tiles[t][0], ite = tiles[t][1], jts = tiles[t][2], jte = tiles[t][3];
        if (its < c->ims || ite > c->ime || jts < c->jms || jte > c->jme) { icar_set_error("thompson: tile outside memory bounds"); return 1; }
        const int i_end = ite < ide - 1 ? ite : ide - 1, j_end = jte < jde - 1 ? jte : jde - 1;
        if (i_end < its || j_end < jts) continue;
        T4[nt_][0] = its; T4[nt_][1] = i_end; T4[nt_][2] = jts; T4[nt_][3] = j_end; ++nt_;
    }
    if (nt_ == 0) return 0;
    ScopedTimer tm(c, "mp");
    // Packed layout (column_comm.h) unless one column per 64-lane wave fills the lanes as well (52 <= nk <= 64).
    int cpb = 0, nt = 0;
    if (nk >= 2) {
        const float u = block_comm_geometry(nk, nt, cpb);
        if (nk <= 64 && u <= nk / 64.0f + 0.02f) { cpb = 0; nt = 0; }
    }
    if (cpb) {
        // all tiles in ONE launch: process_halo's four 1-cell strips are latency-bound when launched one after another
        ThTiles tl; tl.n = nt_; tl.off[0] = 0;
        tl.xcd_run = 64;                 // consecutive column groups (and a few rows of them) per XCD turn
        for (int t = 0; t < nt_; ++t) {
            tl.i0[t] = T4[t][0] - c->ims; tl.i1[t] = T4[t][1] - c->ims; tl.j0[t] = T4[t][2] - c->jms; tl.j1[t] = T4[t][3] - c->jms;
            tl.ib0[t] = tl.i0[t] / cpb; tl.nbx[t] = tl.i1[t] / cpb - tl.ib0[t] + 1;
            const int rows = T4[t][3] - T4[t][2] + 1;
            tl.tall[t] = (tl.i0[t] == tl.i1[t] && rows > 1) ? 1 : 0;
            tl.off[t + 1] = tl.off[t] + (tl.tall[t] ? (rows + cpb - 1) / cpb : tl.nbx[t] * rows);
        }
        if (nt <= 512)
            hipLaunchKernelGGL(k_thompson_pack<512>, dim3(tl.off[nt_]), dim3(nt), BlockComm::lds_bytes(nt, cpb), c->stream, c->d, T,
                               qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, dt, tl, kts - c->kms, nk, cpb);
        else
            hipLaunchKernelGGL(k_thompson_pack<1024>, dim3(tl.off[nt_]), dim3(nt), BlockComm::lds_bytes(nt, cpb), c->stream, c->d, T,
                               qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, dt, tl, kts - c->kms, nk, cpb);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (nk > 64) { icar_set_error("thompson: this many levels are not supported by this build"); return 1; }
    for (int t = 0; t < nt_; ++t) {                      // one column per wave, level = lane
        const int its = T4[t][0], i_end = T4[t][1], jts = T4[t][2], j_end = T4[t][3];
        dim3 gl((i_end - its + 1 + 3) / 4, j_end - jts + 1), bl(256);
        hipLaunchKernelGGL(k_thompson_lane, gl, bl, 0, c->stream, c->d, T, qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga,
                           dt, its - c->ims, i_end - c->ims, jts - c->jms, kts - c->kms, nk);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int icar_thompson_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte,
                      int ids, int ide, int jds, int jde, int kds, int kde)
{
    const int tile[1][4] = {{its, ite, jts, jte}};
    return icar_thompson_run_tiles(c, dt, 1, tile, kts, kte, ids, ide, jds, jde, kds, kde);
}
