// icar_amd/csrc/mp_thompson.hip -- Thompson et al. (2008) bulk microphysics on gfx950 (rows M2/M3).
//
// Reference: src/physics/mp_thompson.f90 -- mp_gt_driver :772-1044 (column gather/scatter, i_end/j_end
// clipping, precipitation accumulation, the qv floor of :997-1010) and mp_thompson :1057-2844 (the
// 1-D column physics).  One column per lane, lanes along i so every level-k access of a
// wave is one coalesced row.  State is REAL(4), rates and lookup-table values REAL(8) exactly as in
// the reference; table indices are integer-exact.  Float transcendentals are evaluated in FP64 and
// rounded once (within 1 ulp of the host libm the reference uses; see tests/test_gpu_thompson.py).
//
// The reference keeps ~130 per-level work arrays.  Here the per-level phases are fused (saturation /
// snow moments / rain slopes / warm rain / frozen processes / conservation / tendencies in one level
// loop; TAU+1 update / condensation / rain evaporation in a second) so that all process rates are
// registers; only the state that crosses levels (graupel N0 chain, fall-speed carry-down,
// sedimentation) stays in lane-interleaved private arrays.
#include "ctx.h"
#include "thompson_state.h"
#include "fp64_math.h"
#define GF_LDS_TABLES          // the look-up tables of expf / logf / powf in LDS: every kernel below starts with gf_lds_init()
#include "glibc_flt32.h"
#define GD_LDS_TABLES          // ... and those of the DOUBLE PRECISION pow / log / exp (7 KB): th_lds_init() = both
#include "glibc_dbl64.h"
__device__ __forceinline__ void th_lds_init(int tid, int nthreads) { gd_lds_init(tid, nthreads); gf_lds_init(tid, nthreads); }
#include "column_comm.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

const ThState *icar_thompson_device_state(icar_hip_ctx *c);
const ThState *icar_thompson_host_state(icar_hip_ctx *c);

namespace {
// DOUBLE PRECISION x**y, log, exp: the C library's pow / log / exp bit for bit (glibc_dbl64.h), which is what the compiled reference
// calls (round 4: until then exp(y log x) with FP64 polynomials of our own -- < 1 ulp of the double, and one float ulp away from
// the reference in ~1e-7 of the cells of a step).  pow is log_inline (a function of the base alone: the powers of one base share
// it, d_plog / d_pow_l -- the same bits as separate pow calls) followed by exp_inline.
// every kernel below holds `const DK K_ = d_consts();` (fp64_math.h) for the REAL(4) helpers that still take it
#define d_exp(x) gd_exp(x)
#define d_log(x) gd_log(x)
#define d_pow(x, y) d_pow_k((x), (y))
#define d_pow_lx(L, x, y) d_pow_lx_k((L), (x), (y))
#define d_plog(x) gd_pow_log(gd_asuint64(x))          /* of a positive, normal DOUBLE PRECISION base */
#define d_powf(x, y) d_powf_k(K_, (x), (y))
#define d_pow_l(L, y) d_pow_l_k((L), (y))
#define d_powf_l(L, y) d_powf_l_k(K_, (L), (y))
#define d_pow10f(y) d_pow10f_k(K_, (y))
#define d_expf(x) d_expf_k(K_, (x))
#define d_log10f(x) d_log10f_k(K_, (x))
// x**y from L = log_inline(x) for the scheme's exponents (finite, 2^-65 <= |y| < 2^63, or zero)
__device__ __forceinline__ double d_pow_l_k(const GdLog &L, double y) { return (y == 0.0) ? 1.0 : gd_pow_exp(L, y, 0); }
// x**1 is x: glibc's pow errs by less than one ulp (0.52), and the only double within one ulp of x is x -- so the library itself
// returns x, bit for bit (checked on 1e9 bases in tests/glibc_dbl64_check.cpp, class pow_one).  The exponents mu_r + 1 and mu_g + 1
// of N0_r / N0_g are 1 with the default parameters: a quarter of a column's pow calls.  The test is wave-uniform (a parameter).
__device__ __forceinline__ double d_pow_k(double x, double y) { return (y == 1.0) ? x : gd_pow(x, y); }
__device__ __forceinline__ double d_pow_lx_k(const GdLog &L, double x, double y) { return (y == 1.0) ? x : d_pow_l_k(L, y); }
// REAL(4) x**y, exp, log10: the C library's powf / expf / log10f bit for bit (glibc_flt32.h), which is what the compiled
// reference calls.  powf is exp2(y * log2 x) with the log2 part a function of the base alone: powers of one base share it
// (PowBase; the same bits as separate powf calls, any base -- an unusual one takes powf itself).
__device__ __forceinline__ float d_powf_k(const DK &, float x, float y) { return gf_powf(x, y); }
struct PowBase { double l2; float x; };
__device__ __forceinline__ PowBase d_powf_base(float x) { PowBase b; b.x = x; b.l2 = gf_powf_log2(gf_asuint(x)); return b; }
__device__ __forceinline__ float d_powf_l_k(const DK &, const PowBase &b, float y) { return gf_powf_from_log2(b.x, b.l2, y); }
// 10.**y (REAL y): powf(10, y) with its log2 part, gf_powf_log2(bits of 10.0f), folded (tests/test_gpu_glibc_math.py op 9)
__device__ __forceinline__ float d_pow10f_k(const DK &, float y) { return gf_powf_from_log2(10.0f, 0x1.a934f0979b22dp+1, y); }
__device__ __forceinline__ float d_expf_k(const DK &, float x) { return gf_expf(x); }
__device__ __forceinline__ float d_log10f_k(const DK &, float x) { return gf_log10f(x); }


/* 10.**nn with an INTEGER exponent: flang calls __powisf2 (repeated squaring) */
__device__ __forceinline__ float powi10f(int b)
{
    const int recip = b < 0;
    float a = 10.0f, r = 1.0f;
    if (recip) b = -b;
    while (1) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return recip ? 1.0f / r : r;
}

/* decade-table index: :1562-1574 and siblings (REAL argument) */
__device__ __forceinline__ int dec_index_f_slow(const DK &K_, float r, int n2)
{
    const int nic = (int)lroundf(d_log10f(r));
    int n = nic - 1;
    for (int nn = nic - 1; nn <= nic + 1; ++nn) {
        n = nn;
        if ((r / powi10f(nn)) >= 1.0f && (r / powi10f(nn)) < 10.0f) break;
    }
    return (int)(r / powi10f(n)) + 10 * (n - n2) - (n - n2);
}

/* same with a DOUBLE PRECISION argument (:1620-1627) */
__device__ __forceinline__ int dec_index_d_slow(const DK &K_, double r, int n2)
{
    const int nic = (int)lround(log10(r));
    int n = nic - 1;
    for (int nn = nic - 1; nn <= nic + 1; ++nn) {
        n = nn;
        if ((r / (double)powi10f(nn)) >= 1.0 && (r / (double)powi10f(nn)) < 10.0) break;
    }
    return (int)(r / (double)powi10f(n)) + 10 * (n - n2) - (n - n2);
}

/* The two routines above cost ~200 instructions per index (a logarithm, up to three trips of repeated squaring, a
 * reciprocal and two divisions each) and a level evaluates up to eight of them.  Their result is n = the decade D with
 * 10**D <= r < 10**(D+1) whenever r is not within rounding distance of a power of ten: the loop starts at nic-1 with
 * nic = nint(log10 r) in {D, D+1}, its test fails for D-1 and holds for D.  So: D from the hardware log2 (error ~1e-5
 * decades); if the fractional part of log10 r is at least 2e-4 away from 0 and 1 (the float powers of ten are within 1e-6
 * of the exact ones) the index is (int)(r / 10**D) + 9 (D - n2) with the SAME float 10**D (table filled by powi10f) and
 * the same IEEE division; otherwise (about 4 values in 10^4) the reference's loop runs. */
__device__ __forceinline__ bool dec_fast(const ThState *__restrict__ T, float rf, int &D, float &p)
{
    const float t = __builtin_amdgcn_logf(rf) * 0.30102999566f;          /* v_log_f32 = log2 */
    const float fl = floorf(t), fr = t - fl;
    D = (int)fl;
    const bool ok = (fr > 2.e-4f) && (fr < 1.0f - 2.e-4f) && (D >= -TH_P10_OFF + 1) && (D <= TH_P10_N - TH_P10_OFF - 2) && (rf > 1.e-37f);
    p = T->p10[ok ? D + TH_P10_OFF : TH_P10_OFF];
    return ok;
}
__device__ __forceinline__ int dec_index_f_k(const DK &K_, const ThState *__restrict__ T, float r, int n2)
{
    int D; float p;
    if (dec_fast(T, r, D, p)) return (int)(r / p) + 10 * (D - n2) - (D - n2);
    return dec_index_f_slow(K_, r, n2);
}
__device__ __forceinline__ int dec_index_d_k(const DK &K_, const ThState *__restrict__ T, double r, int n2)
{
    int D; float p;
    if (r < 1.e37 && dec_fast(T, (float)r, D, p)) return (int)(r / (double)p) + 10 * (D - n2) - (D - n2);
    return dec_index_d_slow(K_, r, n2);
}

#define dec_index_f(T, r, n2) dec_index_f_k(K_, (T), (r), (n2))
#define dec_index_d(T, r, n2) dec_index_d_k(K_, (T), (r), (n2))

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

// cpb whole columns per block (aligned to multiples of cpb in i), thread = level*cpb + column (thompson_lane.inc: BlockComm)
// MAXT = largest block this instantiation is launched with.  Blocks of up to 512 threads (columns of up to 512 levels) get the
// register budget of 3 waves per SIMD (168 VGPRs, no spills); at 128 VGPRs / 4 waves the kernel ran exactly as fast but spilled
// 62 VGPRs -- 4 GB of scratch traffic per launch against 0.9 GB of algorithmic bytes.  1024-thread blocks need the 128.
template <int MAXT>
// Waves per SIMD of the 256-thread launch: until the DOUBLE PRECISION functions became glibc's (table look-ups from LDS: more
// latency to hide, and no scalar registers held for polynomial coefficients any more) three waves at 168 VGPRs beat four at 128
// (1.93 vs 1.98 ms); since then four win -- 1.77 against 1.86 ms alone, 3.10 against 3.15 ms per step at 512 x 512 x 40 (53 VGPRs
// in scratch, no SGPR spills), equal on the small tiles (profiles/r04_steps.md).
#ifndef TH_PACK_WAVES
#define TH_PACK_WAVES 4
#endif
__global__ void __launch_bounds__(MAXT, MAXT > 512 ? 4 : TH_PACK_WAVES)
k_thompson_pack(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                float dt, ThTiles tl, int k0, int nk, int cpb, const IcarDtBlock *__restrict__ blk)
{
    if (blk) dt = blk->mp_dt;                       // graph replay: model_time - last_model_time lives in device memory (timestep.hip)
    extern __shared__ double lds_pack[];
    th_lds_init(threadIdx.x, blockDim.x);
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
    const int j = tall ? (x.active ? first + x.col : first) : tl.j0[t] + local / tl.nbx[t];
    const int i = tall ? i0 : (x.active ? first + x.col : max(i0, min(i1, first)));
    const int c = d.idx(i, k0 + x.k, j);
    const float pi_ = pii[c];
    float t1d = th[c] * pi_, p1d = p[c], dz1d = dz[c], qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c],
          qg1d = qg[c], ni1d = ni[c], nr1d = nr[c];
    float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
    th_column_lane(T, x, nk, dt, dz1d, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, pptrain, pptsnow, pptgraul, pptice);
    if (!x.active) return;
    if (x.k == 0) {
        const int c2 = i + d.nx * j;
        const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
        const float snownc = 0.f + pptsnow + pptice;
        const float graupelnc = 0.f + pptgraul;
        rain_acc[c2] = rain_acc[c2] + rainnc;
        snow_acc[c2] = snow_acc[c2] + snownc;
        graupel_acc[c2] = graupel_acc[c2] + graupelnc;
    }
    qv[c] = (qv1d < 1.E-7f) ? 1.E-7f : qv1d;              // :997-1010
    qc[c] = qc1d; qi[c] = qi1d; qr[c] = qr1d; qs[c] = qs1d; qg[c] = qg1d; ni[c] = ni1d; nr[c] = nr1d;
    th[c] = t1d / pi_;
}

// ---- one column per LANE, levels marched top-down (north_star's layout; round 4) ---------------------------------------------
// A wave owns 64 neighbouring columns (lanes along i) and every VALU instruction works on ONE level of them, so the lanes of a
// wave take the same branches (rain below / ice above no longer share a wave).  Every vertical coupling of the scheme is
// sequential in a thread that walks down its column: the graupel intercept's running minimum (:1456-1468, :2379-2391), the
// fall speeds copied from the nearest level above that holds the species (:2544-2639); only nstep = max over the column of the
// sub-step counts (:2548-2649) needs the whole column before the sedimentation starts.  So two sweeps:
//   sweep 1 (top-down): point physics of each level, ThHand -> coalesced HBM workspace ws[level][value][column]
//   sweep 2 (top-down): sedimentation + melt / freeze + update.  The sub-steps of a level need, for n = 1..nstep, the flux that
//           left the level above in sub-step n (:2660-2770 computes every level's flux before it updates any level): the wave
//           keeps that history in LDS (one 256-B row per sub-step and moment), reads row n, overwrites it with its own flux.
// No barriers, no cross-lane traffic.  If a wave's sub-step counts do not fit its LDS rows the sub-steps are done in chunks,
// one extra sweep over the workspace per chunk.
/* :2660-2770, the sub-steps n of one chunk for one level; `up` = what left the level above in the same sub-step (row r of the
 * species' flux history), overwritten with this level's own flux.  Two-moment (rain, cloud ice) and one-moment (snow, graupel) forms. */
#define TH_SED2(S, HM, HN, VM, VN, QM, QN, TM, TN, FLOORN, PPT)                                                      \
    for (int n = sw * C[S] + 1, r = 0; r < C[S] && n <= nw[S]; ++n, ++r) {                                          \
        if (n <= nstep[S]) {                                                                                         \
            const float sed_m = VM * QM, sed_n = VN * QN;                                                            \
            const float up_m = (k < kte) ? HM[64 * r] : 0.f, up_n = (k < kte) ? HN[64 * r] : 0.f;                    \
            if (k == kte) {                                                                                          \
                TM = TM - sed_m * odzq * onstep[S] * orho;                                                           \
                TN = TN - sed_n * odzq * onstep[S] * orho;                                                           \
                QM = fmaxf(R1, QM - sed_m * odzq * dt * onstep[S]);                                                  \
                QN = fmaxf(FLOORN, QN - sed_n * odzq * dt * onstep[S]);                                              \
            } else if (k <= ksed1[S]) {                                                                              \
                TM = TM + (up_m - sed_m) * odzq * onstep[S] * orho;                                                  \
                TN = TN + (up_n - sed_n) * odzq * onstep[S] * orho;                                                  \
                QM = fmaxf(R1, QM + (up_m - sed_m) * odzq * dt * onstep[S]);                                         \
                QN = fmaxf(FLOORN, QN + (up_n - sed_n) * odzq * dt * onstep[S]);                                     \
            }                                                                                                        \
            if (k == 0 && QM > R1 * 10.f) PPT = PPT + sed_m * dt * onstep[S];                                        \
            HM[64 * r] = sed_m; HN[64 * r] = sed_n;                                                                  \
        }                                                                                                            \
    }
#define TH_SED1(S, HM, VM, QM, TM, PPT)                                                                               \
    for (int n = sw * C[S] + 1, r = 0; r < C[S] && n <= nw[S]; ++n, ++r) {                                          \
        if (n <= nstep[S]) {                                                                                         \
            const float sed_m = VM * QM;                                                                             \
            const float up_m = (k < kte) ? HM[64 * r] : 0.f;                                                         \
            if (k == kte) {                                                                                          \
                TM = TM - sed_m * odzq * onstep[S] * orho;                                                           \
                QM = fmaxf(R1, QM - sed_m * odzq * dt * onstep[S]);                                                  \
            } else if (k <= ksed1[S]) {                                                                              \
                TM = TM + (up_m - sed_m) * odzq * onstep[S] * orho;                                                  \
                QM = fmaxf(R1, QM + (up_m - sed_m) * odzq * dt * onstep[S]);                                         \
            }                                                                                                        \
            if (k == 0 && QM > R1 * 10.f) PPT = PPT + sed_m * dt * onstep[S];                                        \
            HM[64 * r] = sed_m;                                                                                      \
        }                                                                                                            \
    }

struct MarchComm {
    int k; bool active;
    double run_min[2]; float ca[2][4];
    __device__ __forceinline__ MarchComm() : k(0), active(true)
    {
        run_min[0] = run_min[1] = __builtin_inf();
        for (int w = 0; w < 2; ++w) for (int s = 0; s < 4; ++s) ca[w][s] = 0.f;       /* vtXk(kte+1) = 0 */
    }
    __device__ __forceinline__ bool any(bool) { return true; }          /* quiet columns are found by the kernel's first sweep */
    __device__ __forceinline__ double suffix_min(double v, int which) { run_min[which] = fmin(run_min[which], v); return run_min[which]; }
    __device__ __forceinline__ void carry_down2x2(float &a0, float &b0, int has0, float &a1, float &b1, int has1, int which)
    {
        if (has0) { ca[which][0] = a0; ca[which][1] = b0; } else { a0 = ca[which][0]; b0 = ca[which][1]; }
        if (has1) { ca[which][2] = a1; ca[which][3] = b1; } else { a1 = ca[which][2]; b1 = ca[which][3]; }
    }
};

#define TH_MARCH_ROWS 64          /* LDS flux-history rows (256 B each) per wave: 16 KB, 8 waves per CU */

struct MarchArgs { int i0, ni, j0, ncol, ncolp, k0, nk; };

__device__ __forceinline__ int th_wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }

// Register budget: 2 waves per SIMD (<= 256 VGPRs; the kernel takes 224, no scratch).  At 3 waves per SIMD (168 VGPRs) the
// level loop spills ~60 VGPRs on top of ~200 SGPRs parked in VGPR lanes, and that build returned wrong rain numbers at
// the top rain level of some columns (same source; with -DTH_MARCH_WAVES=2 every test is bit-exact) -- icar_amd/build.py
// refuses a build of this kernel that needs scratch.
#ifndef TH_MARCH_WAVES
#define TH_MARCH_WAVES 2
#endif
__global__ void __launch_bounds__(64, TH_MARCH_WAVES)
k_thompson_march(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                 float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                 float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                 double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                 float dt, MarchArgs a, float *__restrict__ ws)
{
    __shared__ float hist[TH_MARCH_ROWS * 64];
    th_lds_init(threadIdx.x, 64);
    const float R1 = TH_R1, R2 = TH_R2, eps = TH_eps;
    const int lane = threadIdx.x, col = blockIdx.x * 64 + lane, nk = a.nk, kte = nk - 1;
    const bool on = col < a.ncol;
    const int colc = on ? col : a.ncol - 1;
    const int jj = a.j0 + colc / a.ni, ii = a.i0 + colc % a.ni;
    const int base = d.idx(ii, a.k0, jj), sk = d.nx;

    /* ---- which columns have nothing to do (:1240-1363): no hydrometeor above R1 and no ice supersaturation at any level ---- */
    bool quiet = true;
    for (int k = 0; k < nk; ++k) {
        const int c = base + k * sk;
        const bool wet = (qc[c] > R1) || (qi[c] > R1) || (qr[c] > R1) || (qs[c] > R1) || (qg[c] > R1);
        const float temp = th[c] * pii[c], pres = p[c], qv_ = fmaxf(1.E-10f, qv[c]);
        const float qvs_ = rslf(pres, temp);
        const float qvsi_ = (temp - 273.15f <= 0.0f) ? rsif(pres, temp) : qvs_;
        float ssati_ = qv_ / qvsi_ - 1.f;
        if (fabsf(ssati_) < eps) ssati_ = 0.0f;
        if (wet || ssati_ > 0.0f) quiet = false;
        if (!__any(quiet && on)) break;
    }
    const bool live = on && !quiet;
    int nstep[4] = {0, 0, 0, 0}, ksed1[4] = {0, 0, 0, 0};
    float onstep[4] = {1.f, 1.f, 1.f, 1.f};

    if (__any(live)) {
        /* ---- sweep 1: point physics, top-down ---- */
        MarchComm x;
        int nsmax[4] = {0, 0, 0, 0}, ks[4] = {-1, -1, -1, -1};
        for (int k = kte; k >= 0; --k) {
            const int c = base + k * sk;
            x.k = k;
            const float pi_ = pii[c];
            float t1d = th[c] * pi_, p1d = p[c], dz1d = dz[c], qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c],
                  qg1d = qg[c], ni1d = ni[c], nr1d = nr[c];
            ThHand h;
            th_level_physics(T, x, nk, dt, dz1d, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, h);
            for (int s = 0; s < 4; ++s) {
                if (h.c4[s] && ks[s] < 0) ks[s] = k;                      /* top-down: the first one is the highest (:2548) */
                nsmax[s] = max(nsmax[s], h.ns4[s]);
            }
            float *w = ws + (size_t)k * TH_NHAND * a.ncolp + col;
            const float hv[TH_NHAND] = {h.vtrk, h.vtnrk, h.vtik, h.vtnik, h.vtsk, h.vtgk, h.rr, h.nr, h.ri, h.ni, h.rs, h.rg,
                                        h.qrten, h.nrten, h.qiten, h.niten, h.qsten, h.qgten, h.qcten, h.qvten, h.tten,
                                        h.rho, h.temp, h.ocp, h.lvap};
            for (int v = 0; v < TH_NHAND; ++v) w[(size_t)v * a.ncolp] = hv[v];
        }
        for (int s = 0; s < 4; ++s) {                                     /* the plan: BlockComm::sed_plan4 for one column */
            const int n = live ? nsmax[s] : 0;
            int kk = ks[s] < 0 ? 0 : ks[s];
            if (kk == kte) kk = kte - 1;
            ksed1[s] = kk; onstep[s] = (n > 0) ? 1.f / (float)n : 1.0f;
            nstep[s] = live ? (int)lroundf(1.f / onstep[s]) : 0;
        }
    }
    /* ---- sweep 2: sedimentation in chunks of sub-steps that fit the LDS rows (one chunk unless nstep is unusually large) ---- */
    int nw[4], C[4];
    for (int s = 0; s < 4; ++s) nw[s] = th_wave_max(nstep[s]);
    if (2 * nw[0] + 2 * nw[1] + nw[2] + nw[3] <= TH_MARCH_ROWS) { for (int s = 0; s < 4; ++s) C[s] = nw[s]; }
    else { for (int s = 0; s < 4; ++s) C[s] = min(nw[s], TH_MARCH_ROWS / 6); }
    int nsweep = 1;
    for (int s = 0; s < 4; ++s) if (C[s] > 0) nsweep = max(nsweep, (nw[s] + C[s] - 1) / C[s]);
    float *H0 = hist + lane, *H1 = H0 + 64 * C[0], *H2 = H1 + 64 * C[0], *H3 = H2 + 64 * C[1], *H4 = H3 + 64 * C[1], *H5 = H4 + 64 * C[2];
    float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
    for (int sw = 0; sw < nsweep; ++sw) {
        const bool last = (sw == nsweep - 1);
        for (int k = kte; k >= 0; --k) {
            const int c = base + k * sk;
            ThHand h;
            if (live) {
                const float *w = ws + (size_t)k * TH_NHAND * a.ncolp + col;
                float hv[TH_NHAND];
                for (int v = 0; v < TH_NHAND; ++v) hv[v] = w[(size_t)v * a.ncolp];
                h.vtrk = hv[0]; h.vtnrk = hv[1]; h.vtik = hv[2]; h.vtnik = hv[3]; h.vtsk = hv[4]; h.vtgk = hv[5];
                h.rr = hv[6]; h.nr = hv[7]; h.ri = hv[8]; h.ni = hv[9]; h.rs = hv[10]; h.rg = hv[11];
                h.qrten = hv[12]; h.nrten = hv[13]; h.qiten = hv[14]; h.niten = hv[15]; h.qsten = hv[16]; h.qgten = hv[17];
                h.qcten = hv[18]; h.qvten = hv[19]; h.tten = hv[20]; h.rho = hv[21]; h.temp = hv[22]; h.ocp = hv[23]; h.lvap = hv[24];
                const float odzq = 1.f / dz[c], orho = 1.f / h.rho;
                /* :2660-2770, sub-steps n of this chunk; `up` = what left the level above in the same sub-step */
                TH_SED2(0, H0, H1, h.vtrk, h.vtnrk, h.rr, h.nr, h.qrten, h.nrten, R2, pptrain)
                TH_SED2(1, H2, H3, h.vtik, h.vtnik, h.ri, h.ni, h.qiten, h.niten, R2, pptice)
                TH_SED1(2, H4, h.vtsk, h.rs, h.qsten, pptsnow)
                TH_SED1(3, H5, h.vtgk, h.rg, h.qgten, pptgraul)
                if (!last) {                                             /* park what the sub-steps moved for the next chunk */
                    float *wr = ws + (size_t)k * TH_NHAND * a.ncolp + col;
                    const float back[12] = {h.rr, h.nr, h.ri, h.ni, h.rs, h.rg, h.qrten, h.nrten, h.qiten, h.niten, h.qsten, h.qgten};
                    for (int v = 0; v < 12; ++v) wr[(size_t)(6 + v) * a.ncolp] = back[v];
                }
            }
            if (last) {
                const float pi_ = pii[c];
                float t1d = th[c] * pi_, qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c], qg1d = qg[c],
                      ni1d = ni[c], nr1d = nr[c];
                /* :1240-1319 what the column routine does to its arguments before anything else */
                if (!(qc1d > R1)) qc1d = 0.0f;
                if (!(qi1d > R1)) { qi1d = 0.0f; ni1d = 0.0f; }
                if (!(qr1d > R1)) { qr1d = 0.0f; nr1d = 0.0f; }
                if (!(qs1d > R1)) qs1d = 0.0f;
                if (!(qg1d > R1)) qg1d = 0.0f;
                if (live) th_level_finish(T, dt, h, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d);
                if (on) {
                    qv[c] = (qv1d < 1.E-7f) ? 1.E-7f : qv1d;          // :997-1010
                    qc[c] = qc1d; qi[c] = qi1d; qr[c] = qr1d; qs[c] = qs1d; qg[c] = qg1d; ni[c] = ni1d; nr[c] = nr1d;
                    th[c] = t1d / pi_;
                }
            }
        }
    }
    if (on) {
        const int c2 = ii + d.nx * jj;
        const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
        const float snownc = 0.f + pptsnow + pptice;
        const float graupelnc = 0.f + pptgraul;
        rain_acc[c2] = rain_acc[c2] + rainnc;
        snow_acc[c2] = snow_acc[c2] + snownc;
        graupel_acc[c2] = graupel_acc[c2] + graupelnc;
    }
}


// ---- 64 columns x 4 levels per block ("slab"), marching down the column in slabs (round 4) ---------------------------------------
// The lane mapping of k_thompson_march (a wave = 64 neighbouring columns of ONE level) with the register budget and the
// parallelism of k_thompson_pack: a 256-thread block owns 64 columns; wave w works on level kte - (4 s + w) of slab s, so a
// thread still holds ONE level at a time (168 VGPRs, 3 waves per SIMD) and four levels of a column advance together.  The
// vertical couplings stay top-down: inside a slab they go through LDS (4 waves that work on ADJACENT levels of the same
// columns: balanced, their barriers are short); what a column carries from slab to slab -- the two running minima of the
// graupel intercept, the fall speeds of the nearest level above, the sub-step counts -- lives in LDS, not in registers.
//   sweep 1  per slab: point physics (:1240-2650), ThHand -> HBM workspace (as k_thompson_march)
//   sweep 2a wave w sediments species w (rain, cloud ice, snow, graupel: independent of each other, :2660-2770) down the whole
//            column, flux history of the level above in LDS rows, final tendencies back to the workspace
//   sweep 2b per slab: melt / freeze / update (:2777-2842) and the stores of the fields
struct SlabLds {
    double sm[4][64];                 // suffix_min exchange of a slab
    double carry_min[2][64];          // running minimum of each chain over the slabs above
    float cdv[4][4][64]; int cdh[4][2][64];
    float carry_vt[2][4][64];         // fall speeds of the nearest level above that holds the species (after its carry-down)
    int ns[4][64], ks[4][64];         // per column: max sub-step count, highest level with a sedimenting particle
    int quiet[64];
    float ppt[4][64];
    float hist[TH_MARCH_ROWS * 64];
};

struct SlabComm {
    SlabLds *L; int w, lane, k; bool active;
    __device__ __forceinline__ bool any(bool) { return true; }
    __device__ __forceinline__ double suffix_min(double v, int which)
    {
        L->sm[w][lane] = v;
        __syncthreads();
        double r = L->carry_min[which][lane];
        for (int ww = 0; ww <= w; ++ww) r = fmin(r, L->sm[ww][lane]);         // wave 0 is the highest level of the slab
        __syncthreads();
        if (w == 3) L->carry_min[which][lane] = r;
        return r;
    }
    __device__ __forceinline__ void carry_down2x2(float &a0, float &b0, int has0, float &a1, float &b1, int has1, int which)
    {
        L->cdv[w][0][lane] = a0; L->cdv[w][1][lane] = b0; L->cdv[w][2][lane] = a1; L->cdv[w][3][lane] = b1;
        L->cdh[w][0][lane] = has0; L->cdh[w][1][lane] = has1;
        __syncthreads();
        if (!has0) {
            int ww = w - 1;
            while (ww >= 0 && !L->cdh[ww][0][lane]) --ww;
            if (ww >= 0) { a0 = L->cdv[ww][0][lane]; b0 = L->cdv[ww][1][lane]; }
            else { a0 = L->carry_vt[which][0][lane]; b0 = L->carry_vt[which][1][lane]; }
        }
        if (!has1) {
            int ww = w - 1;
            while (ww >= 0 && !L->cdh[ww][1][lane]) --ww;
            if (ww >= 0) { a1 = L->cdv[ww][2][lane]; b1 = L->cdv[ww][3][lane]; }
            else { a1 = L->carry_vt[which][2][lane]; b1 = L->carry_vt[which][3][lane]; }
        }
        __syncthreads();
        if (w == 3) { L->carry_vt[which][0][lane] = a0; L->carry_vt[which][1][lane] = b0; L->carry_vt[which][2][lane] = a1; L->carry_vt[which][3][lane] = b1; }
    }
};

// one level of sweep 1 of k_thompson_slab
__device__ __forceinline__ void
th_slab_level(Dims d, const ThState *__restrict__ T, const float *__restrict__ qv, const float *__restrict__ qc, const float *__restrict__ qr,
              const float *__restrict__ qi, const float *__restrict__ qs, const float *__restrict__ qg, const float *__restrict__ ni, const float *__restrict__ nr,
              const float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
              float dt, MarchArgs a, float *__restrict__ ws, SlabLds *L, int k, int w, int lane, int col, int base)
{
    SlabComm x; x.L = L; x.w = w; x.lane = lane;
    x.active = k >= 0; x.k = x.active ? k : 0;
    const int c = base + x.k * d.nx, nk = a.nk;
    const float pi_ = pii[c];
    float t1d = th[c] * pi_, p1d = p[c], dz1d = dz[c], qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c],
          qg1d = qg[c], ni1d = ni[c], nr1d = nr[c];
    ThHand h;
    th_level_physics(T, x, nk, dt, dz1d, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, h);
    if (x.active) {
        for (int sp = 0; sp < 4; ++sp) {
            if (h.c4[sp]) atomicMax(&L->ks[sp][lane], k);                  /* :2548 the highest level with a sedimenting particle */
            if (h.ns4[sp] > 0) atomicMax(&L->ns[sp][lane], h.ns4[sp]);
        }
        /* (wave-uniform row address + the lane's column as a 32-bit offset: per-lane 64-bit addresses of the 25 rows would be
         * hoisted out of the slab loop and cost ~50 VGPRs) */
        float *wrow = ws + (size_t)__builtin_amdgcn_readfirstlane(k) * TH_NHAND * a.ncolp;
        const unsigned ucol = (unsigned)col;
        const float hv[TH_NHAND] = {h.vtrk, h.vtnrk, h.vtik, h.vtnik, h.vtsk, h.vtgk, h.rr, h.nr, h.ri, h.ni, h.rs, h.rg,
                                    h.qrten, h.nrten, h.qiten, h.niten, h.qsten, h.qgten, h.qcten, h.qvten, h.tten,
                                    h.rho, h.temp, h.ocp, h.lvap};
#pragma unroll
        for (int v = 0; v < TH_NHAND; ++v) (wrow + (size_t)v * a.ncolp)[ucol] = hv[v];
    }
}

// sedimentation of ONE species (S = 0 rain, 1 cloud ice, 2 snow, 3 graupel) down a whole column from the workspace of sweep 1:
// its fall speeds, contents and tendencies are read level by level, the final tendencies (and, between chunks of sub-steps,
// the contents) written back.  Returns what reached the ground (pptrain / pptice / pptsnow / pptgraul).
template <int S>
__device__ __forceinline__ float th_sed_species(float *__restrict__ ws, const MarchArgs &a, const float *__restrict__ dz, int base, int sk, int col,
                                                int kte, float dt, const int nstep[4], const int ksed1[4], const float onstep[4],
                                                const int nw[4], const int C[4], float *HM, float *HN)
{
    const float R1 = TH_R1, R2 = TH_R2;
    /* workspace slots (ThHand order): fall speeds (mass, number), contents (mass, number), tendencies (mass, number) */
    constexpr int vV = S == 0 ? 0 : S == 1 ? 2 : S == 2 ? 4 : 5, vVn = S == 0 ? 1 : 3;
    constexpr int vQ = S == 0 ? 6 : S == 1 ? 8 : S == 2 ? 10 : 11, vQn = S == 0 ? 7 : 9;
    constexpr int vT = S == 0 ? 12 : S == 1 ? 14 : S == 2 ? 16 : 17, vTn = S == 0 ? 13 : 15;
    const int nsweep = C[S] > 0 ? (nw[S] + C[S] - 1) / C[S] : 1;
    float ppt = 0.f;
    for (int sw = 0; sw < nsweep; ++sw) {
        const bool last = (sw == nsweep - 1);
        for (int k = kte; k >= 0; --k) {
            float *wp = ws + (size_t)k * TH_NHAND * a.ncolp + col;
            const float odzq = 1.f / dz[base + k * sk], orho = 1.f / wp[(size_t)21 * a.ncolp];
            float vm = wp[(size_t)vV * a.ncolp], qm = wp[(size_t)vQ * a.ncolp], tm = wp[(size_t)vT * a.ncolp];
            if (S < 2) {
                float vn = wp[(size_t)vVn * a.ncolp], qn = wp[(size_t)vQn * a.ncolp], tn = wp[(size_t)vTn * a.ncolp];
                TH_SED2(S, HM, HN, vm, vn, qm, qn, tm, tn, R2, ppt)
                wp[(size_t)vTn * a.ncolp] = tn;
                if (!last) wp[(size_t)vQn * a.ncolp] = qn;
            } else {
                TH_SED1(S, HM, vm, qm, tm, ppt)
            }
            wp[(size_t)vT * a.ncolp] = tm;
            if (!last) wp[(size_t)vQ * a.ncolp] = qm;
        }
    }
    return ppt;
}

__global__ void __launch_bounds__(256, 3)
k_thompson_slab(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                float dt, MarchArgs a, float *__restrict__ ws)
{
    __shared__ SlabLds L;
    th_lds_init(threadIdx.x, 256);
    const float R1 = TH_R1, eps = TH_eps;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = blockIdx.x * 64 + lane, nk = a.nk, kte = nk - 1;
    const int nslab = (nk + 3) / 4;
    const bool on = col < a.ncol;
    const int colc = on ? col : a.ncol - 1;
    const int jj = a.j0 + colc / a.ni, ii = a.i0 + colc % a.ni;
    const int base = d.idx(ii, a.k0, jj), sk = d.nx;
    if (w == 0) {
        L.quiet[lane] = 1;
        for (int c2 = 0; c2 < 2; ++c2) { L.carry_min[c2][lane] = __builtin_inf(); for (int v = 0; v < 4; ++v) L.carry_vt[c2][v][lane] = 0.f; }
        for (int s = 0; s < 4; ++s) { L.ns[s][lane] = 0; L.ks[s][lane] = -1; L.ppt[s][lane] = 0.f; }
    }
    __syncthreads();
    /* ---- which columns have nothing to do (:1240-1363): each wave looks at every fourth level ---- */
    {
        bool quiet = true;
        for (int k = w; k < nk; k += 4) {
            const int c = base + k * sk;
            const bool wet = (qc[c] > R1) || (qi[c] > R1) || (qr[c] > R1) || (qs[c] > R1) || (qg[c] > R1);
            const float temp = th[c] * pii[c], pres = p[c], qv_ = fmaxf(1.E-10f, qv[c]);
            const float qvs_ = rslf(pres, temp);
            const float qvsi_ = (temp - 273.15f <= 0.0f) ? rsif(pres, temp) : qvs_;
            float ssati_ = qv_ / qvsi_ - 1.f;
            if (fabsf(ssati_) < eps) ssati_ = 0.0f;
            if (wet || ssati_ > 0.0f) quiet = false;
            if (!__any(quiet && on)) break;
        }
        if (!quiet) L.quiet[lane] = 0;
    }
    __syncthreads();
    const bool live = on && !L.quiet[lane];
    int nstep[4] = {0, 0, 0, 0}, ksed1[4] = {0, 0, 0, 0};
    float onstep[4] = {1.f, 1.f, 1.f, 1.f};
    const bool any_live = __syncthreads_or(live);

    if (any_live) {
        /* ---- sweep 1: point physics, slab by slab from the model top ---- */
        for (int s = 0; s < nslab; ++s)
            th_slab_level(d, T, qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, dt, a, ws, &L, kte - (4 * s + w), w, lane, col, base);
        __syncthreads();
        for (int sp = 0; sp < 4; ++sp) {                                      /* the plan: BlockComm::sed_plan4 for one column */
            const int n = live ? L.ns[sp][lane] : 0;
            int kk = L.ks[sp][lane] < 0 ? 0 : L.ks[sp][lane];
            if (kk == kte) kk = kte - 1;
            ksed1[sp] = kk; onstep[sp] = (n > 0) ? 1.f / (float)n : 1.0f;
            nstep[sp] = live ? (int)lroundf(1.f / onstep[sp]) : 0;
        }
    }
    /* ---- sweep 2a: wave w sediments species w down the whole column; chunks of sub-steps that fit the LDS rows ---- */
    int nw[4], C[4];
    for (int s = 0; s < 4; ++s) nw[s] = th_wave_max(nstep[s]);
    if (2 * nw[0] + 2 * nw[1] + nw[2] + nw[3] <= TH_MARCH_ROWS) { for (int s = 0; s < 4; ++s) C[s] = nw[s]; }
    else { for (int s = 0; s < 4; ++s) C[s] = min(nw[s], TH_MARCH_ROWS / 6); }
    float *H0 = L.hist + lane, *H1 = H0 + 64 * C[0], *H2 = H1 + 64 * C[0], *H3 = H2 + 64 * C[1], *H4 = H3 + 64 * C[1], *H5 = H4 + 64 * C[2];
    if (any_live && live) {
        float ppt = 0.f;
        if (w == 0) ppt = th_sed_species<0>(ws, a, dz, base, sk, col, kte, dt, nstep, ksed1, onstep, nw, C, H0, H1);
        else if (w == 1) ppt = th_sed_species<1>(ws, a, dz, base, sk, col, kte, dt, nstep, ksed1, onstep, nw, C, H2, H3);
        else if (w == 2) ppt = th_sed_species<2>(ws, a, dz, base, sk, col, kte, dt, nstep, ksed1, onstep, nw, C, H4, H4);
        else ppt = th_sed_species<3>(ws, a, dz, base, sk, col, kte, dt, nstep, ksed1, onstep, nw, C, H5, H5);
        L.ppt[w][lane] = ppt;
    }
    __syncthreads();            /* (block-wide: the tendencies the four waves wrote are read by all of them below) */
    /* ---- sweep 2b: melt / freeze / update, four levels at a time ---- */
    for (int s = 0; s < nslab; ++s) {
        const int k = kte - (4 * s + w);
        if (k < 0) continue;
        const int c = base + k * sk;
        const float pi_ = pii[c];
        float t1d = th[c] * pi_, qv1d = qv[c], qc1d = qc[c], qi1d = qi[c], qr1d = qr[c], qs1d = qs[c], qg1d = qg[c],
              ni1d = ni[c], nr1d = nr[c];
        /* :1240-1319 what the column routine does to its arguments before anything else */
        if (!(qc1d > R1)) qc1d = 0.0f;
        if (!(qi1d > R1)) { qi1d = 0.0f; ni1d = 0.0f; }
        if (!(qr1d > R1)) { qr1d = 0.0f; nr1d = 0.0f; }
        if (!(qs1d > R1)) qs1d = 0.0f;
        if (!(qg1d > R1)) qg1d = 0.0f;
        if (live) {
            const float *wp = ws + (size_t)k * TH_NHAND * a.ncolp + col;
            ThHand h;
            h.qrten = wp[(size_t)12 * a.ncolp]; h.nrten = wp[(size_t)13 * a.ncolp]; h.qiten = wp[(size_t)14 * a.ncolp]; h.niten = wp[(size_t)15 * a.ncolp];
            h.qsten = wp[(size_t)16 * a.ncolp]; h.qgten = wp[(size_t)17 * a.ncolp]; h.qcten = wp[(size_t)18 * a.ncolp]; h.qvten = wp[(size_t)19 * a.ncolp];
            h.tten = wp[(size_t)20 * a.ncolp]; h.rho = wp[(size_t)21 * a.ncolp]; h.temp = wp[(size_t)22 * a.ncolp]; h.ocp = wp[(size_t)23 * a.ncolp];
            h.lvap = wp[(size_t)24 * a.ncolp];
            th_level_finish(T, dt, h, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d);
        }
        if (on) {
            qv[c] = (qv1d < 1.E-7f) ? 1.E-7f : qv1d;          // :997-1010
            qc[c] = qc1d; qi[c] = qi1d; qr[c] = qr1d; qs[c] = qs1d; qg[c] = qg1d; ni[c] = ni1d; nr[c] = nr1d;
            th[c] = t1d / pi_;
        }
    }
    if (w == 0 && on) {
        const float pptrain = L.ppt[0][lane], pptice = L.ppt[1][lane], pptsnow = L.ppt[2][lane], pptgraul = L.ppt[3][lane];
        const int c2 = ii + d.nx * jj;
        const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
        const float snownc = 0.f + pptsnow + pptice;
        const float graupelnc = 0.f + pptgraul;
        rain_acc[c2] = rain_acc[c2] + rainnc;
        snow_acc[c2] = snow_acc[c2] + snownc;
        graupel_acc[c2] = graupel_acc[c2] + graupelnc;
    }
}

// arguments come from the host so that nothing is folded at compile time: the value must be what a level computes at run time
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
// the decade index of the level code for n values: which = 0 the product's form (dec_index_f / dec_index_d with the table), 1 the
// reference's loop alone -- so that a test can compare them value by value (icar_hip_thompson_dec_index)
__global__ void k_thompson_dec_index(const ThState *T, const float *__restrict__ rf, const double *__restrict__ rd, int n, int n2, int which, int *__restrict__ out)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const DK K_ = d_consts();

    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (rf) out[t] = which ? dec_index_f_slow(K_, rf[t], n2) : dec_index_f(T, rf[t], n2);
    else    out[t] = which ? dec_index_d_slow(K_, rd[t], n2) : dec_index_d(T, rd[t], n2);
}
// The transcendentals of the level code on n arguments.  DOUBLE PRECISION sites: op 0 d_log(x), 1 d_exp(x), 2 d_pow(x, y).
// REAL(4) sites (the C library's float functions restated, glibc_flt32.h; arguments narrowed, results widened): 3 powf(x, y),
// 4 expf(x), 5 logf(x), 6 log10f(x), 7 atanf(x), 8 powf through the shared-base form (d_powf_base + d_powf_l), 9 10.**x.
// (x, y) come from the host so that nothing is folded at compile time.
__global__ void k_thompson_math_probe(int op, int n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const DK K_ = d_consts();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double r;
    if (op == 0) r = d_log(x[t]);
    else if (op == 1) r = d_exp(x[t]);
    else if (op == 2) r = d_pow(x[t], y[t]);
    else if (op == 3) r = (double)d_powf((float)x[t], (float)y[t]);
    else if (op == 4) r = (double)d_expf((float)x[t]);
    else if (op == 5) r = (double)gf_logf((float)x[t]);
    else if (op == 6) r = (double)d_log10f((float)x[t]);
    else if (op == 7) r = (double)gf_atanf((float)x[t]);
    else if (op == 8) { const PowBase b = d_powf_base((float)x[t]); r = (double)d_powf_l(b, (float)y[t]); }
    else r = (double)d_pow10f((float)x[t]);
    out[t] = r;
}
}  // namespace

int icar_thompson_math_probe_run(icar_hip_ctx *c, int op, int n, const double *x, const double *y, double *out)
{
    if (op < 0 || op > 9 || ((op == 2 || op == 3 || op == 8) && !y)) { icar_set_error("math_probe: op must be 0..9 (y required for 2, 3, 8)"); return 1; }
    if (n <= 0) return 0;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&dx, sizeof(double) * n)); HIPCHK(hipMalloc(&dout, sizeof(double) * n));
    HIPCHK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    if (y) { HIPCHK(hipMalloc(&dy, sizeof(double) * n)); HIPCHK(hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_thompson_math_probe, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, n, dx, dy, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout); if (dy) (void)hipFree(dy);
    return 0;
}

int icar_thompson_dec_index_run(icar_hip_ctx *c, const float *rf, const double *rd, int n, int n2, int which, int *out)
{
    const ThState *T = icar_thompson_device_state(c);
    if (!T) { icar_set_error("thompson: call icar_hip_thompson_init first"); return 1; }
    if (n <= 0) return 0;
    float *drf = nullptr; double *drd = nullptr; int *dout = nullptr;
    HIPCHK(hipMalloc(&dout, sizeof(int) * n));
    if (rf) { HIPCHK(hipMalloc(&drf, sizeof(float) * n)); HIPCHK(hipMemcpy(drf, rf, sizeof(float) * n, hipMemcpyHostToDevice)); }
    else    { HIPCHK(hipMalloc(&drd, sizeof(double) * n)); HIPCHK(hipMemcpy(drd, rd, sizeof(double) * n, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_thompson_dec_index, dim3((n + 255) / 256), dim3(256), 0, c->stream, T, drf, drd, n, n2, which, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, dout, sizeof(int) * n, hipMemcpyDeviceToHost));
    hipFree(dout); if (drf) hipFree(drf); if (drd) hipFree(drd);
    return 0;
}

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
    // Lanes along i (k_thompson_march = layout 2, k_thompson_slab = layout 3) only on request (icar_hip_thompson_layout): measured on MI355X at
    // 512 x 512 x 40 it executes 23 % fewer VALU instructions at 67 % instead of 61 % active lanes (neighbouring columns of a
    // level still diverge in the +-1 % noise of the benchmark state) but runs 2.40 ms against 1.77 ms for the packed
    // level-per-thread kernel: two waves per SIMD do not hide its instruction-fetch and memory latencies
    // (profiles/r04_thompson_layout.md).
    if (nt_ == 1 && nk >= 2 && (c->th_layout == 2 || c->th_layout == 3)) {
        if (c->dt_dev) c->dt_bad = true;                 // (these layouts take dt by value only: no graph replay)
        const int ni_ = T4[0][1] - T4[0][0] + 1, nj_ = T4[0][3] - T4[0][2] + 1;
        const long ncol = (long)ni_ * nj_;
        {
            MarchArgs a; a.i0 = T4[0][0] - c->ims; a.ni = ni_; a.j0 = T4[0][2] - c->jms; a.ncol = (int)ncol;
            a.ncolp = (int)((ncol + 63) / 64) * 64; a.k0 = kts - c->kms; a.nk = nk;
            const size_t need = (size_t)TH_NHAND * a.ncolp * nk;
            if (c->th_ws_floats < need) {
                if (c->th_ws) { (void)hipFree(c->th_ws); c->th_ws = nullptr; c->th_ws_floats = 0; }
                HIPCHK(hipMalloc(&c->th_ws, need * sizeof(float)));
                c->th_ws_floats = need;
            }
            if (c->th_layout == 3)
                hipLaunchKernelGGL(k_thompson_slab, dim3(a.ncolp / 64), dim3(256), 0, c->stream, c->d, T, qv, qc, qr, qi, qs, qg, ni, nr,
                                   th, pii, p, dz, pa, sa, ga, dt, a, c->th_ws);
            else
                hipLaunchKernelGGL(k_thompson_march, dim3(a.ncolp / 64), dim3(64), 0, c->stream, c->d, T, qv, qc, qr, qi, qs, qg, ni, nr,
                                   th, pii, p, dz, pa, sa, ga, dt, a, c->th_ws);
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
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
                               qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, dt, tl, kts - c->kms, nk, cpb, c->dt_dev);
        else
            hipLaunchKernelGGL(k_thompson_pack<1024>, dim3(tl.off[nt_]), dim3(nt), BlockComm::lds_bytes(nt, cpb), c->stream, c->d, T,
                               qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, dt, tl, kts - c->kms, nk, cpb, c->dt_dev);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (nk > 64) { icar_set_error("thompson: this many levels are not supported by this build"); return 1; }
    if (c->dt_dev) c->dt_bad = true;
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
