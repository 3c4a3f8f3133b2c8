// icar_amd/csrc/mp_wsm6.hip -- WSM6 microphysics (Hong and Lim 2006; src/physics/mp_wsm6.f90), the kMP_WSM6 slot of mp()'s
// dispatch (mp_driver.f90:98-101, :518-550).  SURVEY 8(f) rank 4.
//
// wsm62D (:185-1384) is level-local except for its three semi-Lagrangian falls (rain; snow + graupel with one mass-weighted
// speed; cloud ice) and the surface flux, so one minor time step runs as
//     k_w6_prep   one thread per CELL    :373-381 clamps, cpm, xl (first loop) ; denfac, saturation, Ni, slopes, fall speeds :438-573
//     k_w6_fall   one thread per COLUMN  nislfv_rain_plm (rain, iter = 1) and nislfv_rain_plm6 (snow + graupel, iter = 1) :574-577
//     k_w6_melt   one thread per CELL    back to mixing ratios :578-585, slopes :596, melting of snow / graupel :599-637, ice speed :641-658
//     k_w6_icefall one thread per COLUMN nislfv_rain_plm (cloud ice, iter = 0) :659-667 and the surface sums :672-697
//     k_w6_rates  one thread per CELL    instant melt / freeze :703-759, slopes, the warm and cold process rates :782-1128,
//                                        conservation and update :1136-1318, saturation adjustment :1330-1385
// with REAL(4) work fields in between (21 fields of the tile: 0.9 GB at 512 x 512 x 40).  Every statement keeps the reference's
// operation order.  REAL(4) exp / log / x**y are the C library's expf / logf / powf bit for bit (glibc_flt32.h), sqrt and divide
// IEEE: oracle/wsm6_oracle.c (a separate restatement, pinned to the compiled reference) calls the host's libm, and
// tests/test_gpu_wsm6.py compares bit for bit.
#include "ctx.h"
#include "glibc_flt32.h"
#include <cmath>
#include <cstring>
#include <cstdlib>

#define W6_MAXK 64

// module parameters mp_wsm6.f90:16-43
#define W6_dtcldcr 120.f
#define W6_n0r 8.e6f
#define W6_n0g 4.e6f
#define W6_avtr 841.9f
#define W6_bvtr 0.8f
#define W6_r0 .8e-5f
#define W6_peaut .55f
#define W6_xncr 3.e8f
#define W6_xmyu 1.718e-5f
#define W6_avts 11.72f
#define W6_bvts .41f
#define W6_avtg 330.f
#define W6_bvtg 0.8f
#define W6_deng 500.f
#define W6_n0smax 1.e11f
#define W6_lamdarmax 8.e4f
#define W6_lamdasmax 1.e5f
#define W6_lamdagmax 6.e4f
#define W6_dicon 11.9f
#define W6_dimax 500.e-6f
#define W6_n0s 2.e6f
#define W6_alpha .12f
#define W6_pfrz1 100.f
#define W6_pfrz2 0.66f
#define W6_qcrmin 1.e-9f
#define W6_eacrc 1.0f
#define W6_dens 100.0f
#define W6_qs0 6.e-4f

namespace {
// what wsm6init derives (:1432-1506); only the ones the kernels read
struct W6Consts {
    float qc0, qck1, pvtr, pacrr, g6pbr, precr1, precr2, roqimax, pvts, precs1, precs2, pidn0r, pidn0s, xlv1, pacrc, pi,
          pvtg, pacrg, precg1, precg2, pidn0g;
    float smax[3], sbmax[3], s2max[3], s3max[3];          // rslope?max, rslope?bmax, rslope?2max, rslope?3max of rain, snow, graupel
};
// what mp_driver.f90:518-550 passes
struct W6Args { float delt, g, cpd, cpv, rd, rv, t0c, ep1, ep2, qmin, xls, xlv0, xlf0, den0, denr, cliq, cice, psat; };
// saturation coefficients of the inlined fpvs (:451-461)
struct W6Sat { float ttp, xa, xb, xai, xbi; };

// REAL(4) exp / log / x**y as the compiled reference evaluates them: the C library's expf / logf / powf (glibc_flt32.h)
__device__ __forceinline__ float e6(float x) { return gf_expf(x); }
__device__ __forceinline__ float l6(float x) { return gf_logf(x); }
__device__ __forceinline__ float p6(float x, float y) { return gf_powf(x, y); }
__device__ __forceinline__ float mx(float a, float b) { return a > b ? a : b; }        // Fortran max / min of two reals
__device__ __forceinline__ float mn(float a, float b) { return a < b ? a : b; }

// one species (s = 0 rain, 1 snow, 2 graupel) of one level: slope_rain / slope_snow / slope_graup (:1585-1720) == the three
// blocks of slope_wsm6 (:1508-1583).  Returns the fall speed.
struct Slope { float r, rb, r2, r3, vt; };
template <int S>
__device__ __forceinline__ Slope w6_slope(const W6Consts &C, float q, float den, float denfac, float t)
{
    Slope o;
    if (q <= W6_qcrmin) { o.r = C.smax[S]; o.rb = C.sbmax[S]; o.r2 = C.s2max[S]; o.r3 = C.s3max[S]; }
    else {
        float lam;
        if (S == 0) lam = sqrtf(sqrtf(C.pidn0r / (q * den)));
        else if (S == 1) {
            const float supcol = 273.15f - t;
            const float n0sfac = mx(mn(e6(W6_alpha * supcol), W6_n0smax / W6_n0s), 1.f);
            lam = sqrtf(sqrtf(C.pidn0s * n0sfac / (q * den)));
        } else lam = sqrtf(sqrtf(C.pidn0g / (q * den)));
        o.r = 1.f / lam;
        o.rb = p6(o.r, S == 0 ? W6_bvtr : S == 1 ? W6_bvts : W6_bvtg);
        o.r2 = o.r * o.r;
        o.r3 = o.r2 * o.r;
    }
    o.vt = (S == 0 ? C.pvtr : S == 1 ? C.pvts : C.pvtg) * o.rb * denfac;
    if (q <= 0.0f) o.vt = 0.0f;
    return o;
}

// statement functions :352-366
#define W6_DIFFUS(x, y) (8.794e-5f * e6(l6(x) * (1.81f)) / (y))
#define W6_VISCOS(x, y) (1.496e-6f * ((x) * sqrtf(x)) / ((x) + 120.f) / (y))
#define W6_XKA(x, y) (1.414e3f * W6_VISCOS(x, y) * (y))
#define W6_DIFFAC(a, b, c, d, e) ((d) * (a) * (a) / (W6_XKA(c, d) * A.rv * (c) * (c)) + 1.f / ((e) * W6_DIFFUS(c, b)))
#define W6_VENFAC(a, b, c) (e6(l6((W6_VISCOS(b, c) / W6_DIFFUS(b, a))) * ((.3333333f))) / sqrtf(W6_VISCOS(b, c)) * sqrtf(sqrtf(A.den0 / (c))))

__device__ __forceinline__ W6Sat w6_sat_coeffs(const W6Args &A)
{
    W6Sat S;
    S.ttp = A.t0c + 0.01f;
    const float dldt = A.cpv - A.cliq; S.xa = -dldt / A.rv; S.xb = S.xa + A.xlv0 / (A.rv * S.ttp);
    const float dldti = A.cpv - A.cice; S.xai = -dldti / A.rv; S.xbi = S.xai + A.xls / (A.rv * S.ttp);
    return S;
}
// saturation mixing ratio over water (ICE = false) / over ice below the triple point (ICE = true): :462-476, :1341-1355
template <bool ICE>
__device__ __forceinline__ float w6_qsat(const W6Args &A, const W6Sat &S, float t, float p)
{
    const float tr = S.ttp / t;
    float v;
    if (ICE && t < S.ttp) v = A.psat * e6(l6(tr) * (S.xai)) * e6(S.xbi * (1.f - tr));
    else v = A.psat * e6(l6(tr) * (S.xa)) * e6(S.xb * (1.f - tr));
    v = mn(v, 0.99f * p);
    v = A.ep2 * v / (p - v);
    return mx(v, A.qmin);
}
__device__ __forceinline__ float w6_xni(const W6Args &A, float den, float qi)                  // :534-540, :872-874
{
    float temp = (den * mx(qi, A.qmin));
    temp = sqrtf(sqrtf(temp * temp * temp));
    return mn(mx(5.38e7f * temp, 1.e3f), 1.e6f);
}

// ---------------- the semi-Lagrangian fall of one column (nislfv_rain_plm :1723-1961, nislfv_rain_plm6 :1963-2230) ----------------
// Column arrays are addressed with the element stride st (level k of a column is k*st away in the (i,k,j) fields).
struct FallGeom { float zi[W6_MAXK + 1], za[W6_MAXK + 1], dza[W6_MAXK + 1]; };

// interface speeds (third-order interpolation, rain-shaft top, the 5 % deformation limiter) and arrival heights :1755-1790
__device__ void w6_arrival(int km, int st, const float *ww, const float *__restrict__ dz, float dt, FallGeom &G)
{
    float wi[W6_MAXK + 1];
    const float fa1 = 9.f / 16.f, fa2 = 1.f / 16.f, con1 = 0.05f;
    wi[0] = ww[0];
    wi[1] = 0.5f * (ww[1] + ww[0]);
    for (int k = 2; k < km - 1; ++k) wi[k] = fa1 * (ww[k] + ww[k - 1]) - fa2 * (ww[k + 1] + ww[k - 2]);
    wi[km - 1] = 0.5f * (ww[km - 1] + ww[km - 2]);
    wi[km] = ww[km - 1];
    for (int k = 1; k < km; ++k) if (ww[k] == 0.0f) wi[k] = ww[k - 1];
    for (int k = km - 1; k >= 0; --k) {
        const float dzk = dz[k * st];
        const float decfl = (wi[k + 1] - wi[k]) * dt / dzk;
        if (decfl > con1) wi[k] = wi[k + 1] - con1 * dzk / dt;
    }
    for (int k = 0; k <= km; ++k) G.za[k] = G.zi[k] - wi[k] * dt;
    for (int k = 0; k < km; ++k) G.dza[k] = G.za[k + 1] - G.za[k];
    G.dza[km] = G.zi[km] - G.za[km];
}

// piecewise-linear reconstruction of qa on the arrival grid, remap onto the regular levels (written to out, stride st), and the
// part that left through the ground (returned): :1815-1948
__device__ float w6_remap(int km, int st, const FallGeom &G, const float *qa, float *__restrict__ out)
{
    float qmi[W6_MAXK + 1], qpi[W6_MAXK + 1];
    const float *zi = G.zi, *za = G.za, *dza = G.dza;
    for (int k = 1; k < km; ++k) {
        const float dip = (qa[k + 1] - qa[k]) / (dza[k + 1] + dza[k]);
        const float dim = (qa[k] - qa[k - 1]) / (dza[k - 1] + dza[k]);
        if (dip * dim <= 0.0f) { qmi[k] = qa[k]; qpi[k] = qa[k]; }
        else {
            qpi[k] = qa[k] + 0.5f * (dip + dim) * dza[k];
            qmi[k] = 2.0f * qa[k] - qpi[k];
            if (qpi[k] < 0.0f || qmi[k] < 0.0f) { qpi[k] = qa[k]; qmi[k] = qa[k]; }
        }
    }
    qpi[0] = qa[0]; qmi[0] = qa[0]; qmi[km] = qa[km]; qpi[km] = qa[km];
    int kb = 1, kt = 1, k = 1;                               // 1-based like the reference's; levels the loop leaves early stay 0
    for (; k <= km; ++k) {
        kb = kb - 1 > 1 ? kb - 1 : 1;
        kt = kt - 1 > 1 ? kt - 1 : 1;
        if (zi[k - 1] >= za[km]) break;
        for (int kk = kb; kk <= km; ++kk) if (zi[k - 1] <= za[kk]) { kb = kk; break; }
        for (int kk = kt; kk <= km; ++kk) if (zi[k] <= za[kk - 1]) { kt = kk; break; }
        kt = kt - 1;
        float qn = 0.0f;
        if (kt == kb) {
            const float tl = (zi[k - 1] - za[kb - 1]) / dza[kb - 1];
            const float th = (zi[k] - za[kb - 1]) / dza[kb - 1];
            const float tl2 = tl * tl, th2 = th * th;
            const float qqd = 0.5f * (qpi[kb - 1] - qmi[kb - 1]);
            const float qqh = qqd * th2 + qmi[kb - 1] * th;
            const float qql = qqd * tl2 + qmi[kb - 1] * tl;
            qn = (qqh - qql) / (th - tl);
        } else if (kt > kb) {
            const float tl = (zi[k - 1] - za[kb - 1]) / dza[kb - 1];
            const float tl2 = tl * tl;
            float qqd = 0.5f * (qpi[kb - 1] - qmi[kb - 1]);
            const float qql = qqd * tl2 + qmi[kb - 1] * tl;
            const float dql = qa[kb - 1] - qql;
            float zsum = (1.f - tl) * dza[kb - 1];
            float qsum = dql * dza[kb - 1];
            for (int m = kb + 1; m <= kt - 1; ++m) { zsum = zsum + dza[m - 1]; qsum = qsum + qa[m - 1] * dza[m - 1]; }
            const float th = (zi[k] - za[kt - 1]) / dza[kt - 1];
            const float th2 = th * th;
            qqd = 0.5f * (qpi[kt - 1] - qmi[kt - 1]);
            const float dqh = qqd * th2 + qmi[kt - 1] * th;
            zsum = zsum + th * dza[kt - 1];
            qsum = qsum + dqh * dza[kt - 1];
            qn = qsum / zsum;
        }
        out[(k - 1) * st] = qn;
    }
    for (; k <= km; ++k) out[(k - 1) * st] = 0.0f;
    float precip = 0.0f;
    for (int kk = 0; kk < km; ++kk) {
        if (za[kk] < 0.0f && za[kk + 1] < 0.0f) { precip = precip + qa[kk] * dza[kk]; continue; }
        else if (za[kk] < 0.0f && za[kk + 1] >= 0.0f) { precip = precip + qa[kk] * (0.0f - za[kk]); break; }
        break;
    }
    return precip;
}

// MODE 0: rain (iter = 1, slope_rain) ; 1: cloud ice (iter = 0) ; 2: snow + graupel (iter = 1, mass-weighted slope_snow / slope_graup)
// rql (and rql2 for MODE 2) hold den*q on input and output.  precip[0..1] = what left through the ground.
template <int MODE>
__device__ void w6_fall_column(const W6Consts &C, int km, int st, const float *__restrict__ den, const float *__restrict__ denfac,
                               const float *__restrict__ tk, const float *__restrict__ dz, const float *__restrict__ wwl,
                               float *__restrict__ rql, float *__restrict__ rql2, float dt, float *precip)
{
    FallGeom G;
    float ww[W6_MAXK], qa[W6_MAXK + 1], qa2[MODE == 2 ? W6_MAXK + 1 : 1];
    precip[0] = 0.0f; precip[1] = 0.0f;
    float allold = 0.0f;
    for (int k = 0; k < km; ++k) {
        ww[k] = wwl[k * st];
        if (MODE == 2) allold = allold + rql[k * st] + rql2[k * st]; else allold = allold + rql[k * st];
    }
    if (allold <= 0.0f) return;                              // cycle i_loop: the column keeps its den*q
    G.zi[0] = 0.0f;
    for (int k = 0; k < km; ++k) G.zi[k + 1] = G.zi[k] + dz[k * st];
    for (int n = 1;; ++n) {
        w6_arrival(km, st, ww, dz, dt, G);
        for (int k = 0; k < km; ++k) {
            qa[k] = rql[k * st] * dz[k * st] / G.dza[k];
            if (MODE == 2) qa2[k] = rql2[k * st] * dz[k * st] / G.dza[k];
        }
        qa[km] = 0.0f;
        if (MODE == 2) qa2[km] = 0.0f;
        if (MODE == 1 || n > 1) break;
        for (int k = 0; k < km; ++k) {                       // one refinement of the speed with the arrived mixing ratios
            const float dk = den[k * st], df = denfac[k * st];
            float wa;
            if (MODE == 0) wa = w6_slope<0>(C, qa[k] / dk, dk, df, 0.f).vt;
            else {
                const float qr = qa[k] / dk, qr2 = qa2[k] / dk;
                const float was = w6_slope<1>(C, qr, dk, df, tk[k * st]).vt, wag = w6_slope<2>(C, qr2, dk, df, 0.f).vt;
                const float tmp = mx(qr + qr2, 1.E-15f);
                if (tmp > 1.e-15f) wa = (was * qr + wag * qr2) / tmp; else wa = 0.f;
            }
            ww[k] = 0.5f * (wwl[k * st] + wa);
        }
    }
    precip[0] = w6_remap(km, st, G, qa, rql);
    if (MODE == 2) precip[1] = w6_remap(km, st, G, qa2, rql2);
}


// ---------------- the same falls with one WAVE per column: lane = level (cells) / interface (wi, zi, za, dza, qa live on lanes 0..km).
// nislfv_rain_plm / _plm6 are sequential in k in four places only, which stay sequential so that every sum and comparison sees the
// reference's operands (the scheme of mp_wsm3.hip's fall, here with DPP wave shifts for the nearest-neighbour reads):
//   zi          the running sum of dz: a field filled once per call (k_w6_zi)
//   wi limiter  k = km..1 uses the wi(k+1) it may just have changed: evaluated for all k at once with the unmodified values;
//               only from the highest level that trips the limit downward is it re-walked serially (rare)
//   kb / kt     "first kk >= previous-1 with zi <= za(kk)": za increases strictly (the limiter guarantees dza >= 0.95 dz), so the
//               first kk is the count of arrival heights below zi -- a binary search per lane; where kt is not found the
//               reference's stale kt is < kb and the level gets qn = 0 either way
//   sums        the kb+1..kt-1 partial sums and the surface flux are short loops in k order
// Needs km + 1 <= 64 lanes; every cross-lane read happens with all lanes active.
__device__ __forceinline__ float w6_up(float x)      // value of lane-1 (0 in lane 0): v_mov_b32_dpp wave_shr:1
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float w6_dn(float x)      // value of lane+1 (0 in lane 63): wave_shl:1
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true)); }

// interface speeds, limiter, arrival heights (:1755-1790): ww per cell lane -> za, dza per interface lane
__device__ __forceinline__ void w6w_arrival(int km, int lane, float ww, float dz, float zi, float dt, float &za, float &dza)
{
    const bool cell = lane < km;
    const float wm1 = w6_up(ww), wm2 = w6_up(wm1), wp1 = w6_dn(ww);
    const float fa1 = 9.f / 16.f, fa2 = 1.f / 16.f, con1 = 0.05f;
    float wi;
    if (lane == 0) wi = ww;
    else if (lane == 1) wi = 0.5f * (ww + wm1);
    else if (lane <= km - 2) wi = fa1 * (ww + wm1) - fa2 * (wp1 + wm2);
    else if (lane == km - 1) wi = 0.5f * (ww + wm1);
    else wi = wm1;                                           // lane == km: wi(km+1) = ww(km)
    if (lane >= 1 && lane < km && ww == 0.0f) wi = wm1;      // terminate at the top of the rain shaft
    const float wip1 = w6_dn(wi);
    const float dec = (wip1 - wi) * dt / dz;
    const unsigned long long bad = __ballot(cell && dec > con1);
    if (bad) {                                               // wave-uniform
        // level k must be re-evaluated when wi(k+1) has just been changed; when a level is left alone everything below it still
        // sees the values the parallel evaluation saw, so the walk jumps to the next level that tripped there
        const float cdz = con1 * dz / dt;
        unsigned long long rem = bad;
        int k = 63 - __builtin_clzll(rem);
        while (k >= 0) {
            const float wk1 = __shfl(wi, k + 1), wk = __shfl(wi, k), dzk = __shfl(dz, k), ck = __shfl(cdz, k);
            const float decfl = (wk1 - wk) * dt / dzk;
            rem &= (k == 0) ? 0ull : ((1ull << k) - 1ull);
            if (decfl > con1) { if (lane == k) wi = wk1 - ck; k = k - 1; }
            else k = rem ? 63 - __builtin_clzll(rem) : -1;
        }
    }
    za = zi - wi * dt;                                       // interfaces 0..km
    const float zap1 = w6_dn(za);
    dza = (lane < km) ? zap1 - za : zi - za;                 // dza(km+1) = zi(km+1) - za(km+1)
}

// reconstruction, remap and rain-out of one arrived field (:1815-1948): returns this lane's qn, adds to precip
__device__ __forceinline__ float w6w_remap(int km, int lane, float zi, float za, float dza, float qa, float &precip)
{
    const bool cell = lane < km;
    float qmi = qa, qpi = qa;
    {
        const float qap1 = w6_dn(qa), qam1 = w6_up(qa), dzap1 = w6_dn(dza), dzam1 = w6_up(dza);
        if (lane >= 1 && lane < km) {
            const float dip = (qap1 - qa) / (dzap1 + dza);
            const float dim = (qa - qam1) / (dzam1 + dza);
            if (!(dip * dim <= 0.0f)) {
                qpi = qa + 0.5f * (dip + dim) * dza;
                qmi = 2.0f * qa - qpi;
                if (qpi < 0.0f || qmi < 0.0f) { qpi = qa; qmi = qa; }
            }
        }
    }
    const float zlo = zi, zhi = w6_dn(zi);                   // the output cell of this lane is [zi(lane), zi(lane+1)]
    const float za_top = __shfl(za, km);
    int lo1 = 0, hi1 = km + 1, lo2 = 0, hi2 = km;            // arrival heights below zlo among 1..km (nb), below zhi among 0..km-1 (nt)
    for (int step = 0; step < 6; ++step) {
        const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
        const float v1 = __shfl(za, m1 < 63 ? m1 : 63), v2 = __shfl(za, m2 < 63 ? m2 : 63);
        if (lo1 < hi1) { if (v1 < zlo) lo1 = m1 + 1; else hi1 = m1; }
        if (lo2 < hi2) { if (v2 < zhi) lo2 = m2 + 1; else hi2 = m2; }
    }
    const float za0 = __shfl(za, 0);
    const int nb = lo1 - (za0 < zlo ? 1 : 0), nt = lo2;
    const bool live = cell && !(zlo >= za_top);              // not yet `exit intp`
    const int kb = live ? nb + 1 : 1;
    const bool found = live && nt < km;
    const int kt = found ? nt : 0;
    const int ib = kb - 1, it = (kt >= 1 ? kt : 1) - 1;
    const float za_b = __shfl(za, ib), dza_b = __shfl(dza, ib), qpi_b = __shfl(qpi, ib), qmi_b = __shfl(qmi, ib), qa_b = __shfl(qa, ib);
    const float za_t = __shfl(za, it), dza_t = __shfl(dza, it), qpi_t = __shfl(qpi, it), qmi_t = __shfl(qmi, it);
    const float tl = (zlo - za_b) / dza_b;
    const float tl2 = tl * tl;
    const float qqd_b = 0.5f * (qpi_b - qmi_b);
    const float qql = qqd_b * tl2 + qmi_b * tl;
    float zsum = (1.f - tl) * dza_b, qsum = (qa_b - qql) * dza_b;
    const int cnt = (found && kt > kb) ? kt - kb - 1 : 0;
    int cmax = cnt;
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(cmax, o); cmax = v > cmax ? v : cmax; }
    for (int s2 = 1; s2 <= cmax; ++s2) {
        const int m = kb + s2 - 1 <= 63 ? kb + s2 - 1 : 63;
        const float dm = __shfl(dza, m), qm = __shfl(qa, m);
        if (s2 <= cnt) { zsum = zsum + dm; qsum = qsum + qm * dm; }
    }
    float qn = 0.0f;
    if (found && kt == kb) {
        const float th = (zhi - za_b) / dza_b;
        const float th2 = th * th;
        const float qqh = qqd_b * th2 + qmi_b * th;
        qn = (qqh - qql) / (th - tl);
    } else if (found && kt > kb) {
        const float th = (zhi - za_t) / dza_t;
        const float th2 = th * th;
        const float qqd = 0.5f * (qpi_t - qmi_t);
        const float dqh = qqd * th2 + qmi_t * th;
        zsum = zsum + th * dza_t;
        qsum = qsum + dqh * dza_t;
        qn = qsum / zsum;
    }
    for (int k = 0; k < km; ++k) {                           // rain out, k ascending (wave-uniform loop on broadcast values)
        const float zk = __shfl(za, k), zk1 = __shfl(za, k + 1), qk = __shfl(qa, k), dk = __shfl(dza, k);
        if (zk < 0.0f && zk1 < 0.0f) { precip = precip + qk * dk; continue; }
        else if (zk < 0.0f && zk1 >= 0.0f) { precip = precip + qk * (0.0f - zk); break; }
        break;
    }
    return qn;
}

// one column on one wave.  MODE as in w6_fall_column.  qn / qn2 = the fallen den*q of this lane's level.
template <int MODE>
__device__ __forceinline__ void w6_fall_wave(const W6Consts &C, int km, int lane, float dz, float den, float denfac, float tk, float wwl,
                                             float rql, float rql2, float zi, float dt, float &qn, float &qn2, float precip[2])
{
    const bool cell = lane < km;
    precip[0] = 0.0f; precip[1] = 0.0f;
    qn = rql; qn2 = rql2;                                    // an empty column keeps den*q as it is (cycle i_loop)
    // allold > 0: den*q >= 0, so the sum is positive iff one term is
    if (__ballot(cell && (rql > 0.0f || (MODE == 2 && rql2 > 0.0f))) == 0ull) return;
    float ww = cell ? wwl : 0.0f, za, dza, qa, qa2 = 0.0f;
    for (int n = 1;; ++n) {
        w6w_arrival(km, lane, ww, dz, zi, dt, za, dza);
        qa = cell ? rql * dz / dza : 0.0f;                   // qa(km+1) = 0
        if (MODE == 2) qa2 = cell ? rql2 * dz / dza : 0.0f;
        if (MODE == 1 || n > 1) break;
        float wa;                                            // one refinement of the speed with the arrived mixing ratios
        if (MODE == 0) wa = w6_slope<0>(C, cell ? qa / den : 0.f, den, denfac, 0.f).vt;
        else {
            const float qr = cell ? qa / den : 0.f, qr2 = cell ? qa2 / den : 0.f;
            const float was = w6_slope<1>(C, qr, den, denfac, tk).vt, wag = w6_slope<2>(C, qr2, den, denfac, 0.f).vt;
            const float tmp = mx(qr + qr2, 1.E-15f);
            if (tmp > 1.e-15f) wa = (was * qr + wag * qr2) / tmp; else wa = 0.f;
        }
        ww = cell ? 0.5f * (wwl + wa) : 0.0f;
    }
    qn = w6w_remap(km, lane, zi, za, dza, qa, precip[0]);
    if (MODE == 2) qn2 = w6w_remap(km, lane, zi, za, dza, qa2, precip[1]);
}

// Up to 4 tiles per launch (the strips of process_halo, mp_driver.f90:609-658, or the one tile of a plain call).  Threads map to the
// cells / columns of the tiles flattened tile after tile, i fastest: full waves whatever a tile's shape (the strips are one
// column wide), coalesced rows for ordinary tiles.
struct W6Tiles { int n, i0[4], i1[4], j0[4], nrow[4], coff[5], boff[5]; };       // coff: prefix of columns, boff: of fall-tile blocks
struct W6Box { int i0, i1, j0, nrow; long long local; };
// tile of flat index t (in units of `unit` per column): constant indices only, so the table stays in SGPRs
__device__ __forceinline__ W6Box w6_box(const W6Tiles &tl, long long t, long long unit, const int *off)
{
    W6Box b = {tl.i0[0], tl.i1[0], tl.j0[0], tl.nrow[0], t};
#pragma unroll
    for (int tt = 1; tt < 4; ++tt)
        if (tt < tl.n && t >= (long long)off[tt] * unit) { b.i0 = tl.i0[tt]; b.i1 = tl.i1[tt]; b.j0 = tl.j0[tt]; b.nrow = tl.nrow[tt]; b.local = t - (long long)off[tt] * unit; }
    return b;
}
#define W6_CELL_INDEX                                                                                   \
    const long long t_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;                               \
    if (t_ >= (long long)tl.coff[tl.n] * km) return;                                                    \
    const W6Box bx_ = w6_box(tl, t_, km, tl.coff);                                                      \
    const int w_ = bx_.i1 - bx_.i0 + 1;                                                                 \
    const int i = bx_.i0 + (int)(bx_.local % w_), k = k0 + (int)((bx_.local / w_) % km), j = bx_.j0 + (int)(bx_.local / ((long long)w_ * km));
#define W6_COLUMN_INDEX                                                                                 \
    const long long t_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;                               \
    if (t_ >= tl.coff[tl.n]) return;                                                                    \
    const W6Box bx_ = w6_box(tl, t_, 1, tl.coff);                                                       \
    const int w_ = bx_.i1 - bx_.i0 + 1;                                                                 \
    const int i = bx_.i0 + (int)(bx_.local % w_), j = bx_.j0 + (int)(bx_.local / w_);

// ---------------- work fields ----------------
struct W6Work {
    float *t, *cpm, *xl, *denfac, *qs1, *qs2, *rh1, *rh2, *xni, *workr, *worka, *dq1, *dq2, *dq3, *vti, *dqi, *frz, *zi;   // (nx, nz, ny)
    float *rain, *snow, *graupel, *delq;                                                                                 // (nx, ny) ; delq: 3 of them
};
}  // namespace

struct Wsm6State {
    W6Consts c; bool ready = false;
    W6Work w = {};
    size_t n3 = 0;
};

namespace {

// zi(k+1) = zi(k) + dz(k) (:1747-1750), the reference's running sum, once per call and column
__global__ void __launch_bounds__(64)
k_w6_zi(Dims d, const float *__restrict__ delz, float *__restrict__ zi, W6Tiles tl, int k0, int km)
{
    W6_COLUMN_INDEX
    float run = 0.0f;
    for (int k = 0; k < km; ++k) { const int c = d.idx(i, k0 + k, j); run = run + delz[c]; zi[c] = run; }
}

// The falls with lane = level.  A block = W6_NT / 64 waves = one row segment of W6_TC columns of one fall (blockIdx.z: 0 rain, 1 snow +
// graupel, 2 cloud ice -- launched as z = 0..1 before the melting kernel and z = 2 after it): the column arrays are staged through
// LDS as [level][column] tiles (coalesced row reads; the column-per-wave access pattern itself would touch one cache line per
// lane), each wave then walks its W6_TC / 4 columns, and the results go back the same way.
// (tile width / block size measured at 512x512x40: 16 / 256 4.03 ms per step; 32 / 256 4.56; 32 / 512 4.28; 8 / 256 4.17; 64 / 512 5.53 --
// the 21 kB of LDS of a 16-column tile keep 7 blocks per CU resident)
#define W6_TC 16
#define W6_NT 256
__global__ void __launch_bounds__(W6_NT)
k_w6_fall_tile(Dims d, W6Consts C, W6Work W, const float *__restrict__ den_, const float *__restrict__ delz, float dt,
               W6Tiles tl, int k0, int km, int zbase)
{
    extern __shared__ float w6_lds[];                                // [8][km][W6_TC + 1]
    const int fall = zbase + blockIdx.z;                             // 0 rain, 1 snow + graupel, 2 cloud ice
    const W6Box bx_ = w6_box(tl, blockIdx.x, 1, tl.boff);            // blocks are numbered tile after tile, x-tile fastest
    const int nxt = (bx_.i1 - bx_.i0 + W6_TC) / W6_TC;
    const int ib = bx_.i0 + (int)(bx_.local % nxt) * W6_TC, j = bx_.j0 + (int)(bx_.local / nxt);
    const int i1 = bx_.i1;
    const int ncol = min(W6_TC, i1 - ib + 1);
    const int LS = W6_TC + 1, plane = km * LS;
    float *__restrict__ dq = fall == 0 ? W.dq1 : fall == 1 ? W.dq2 : W.dqi;
    float *__restrict__ dqb = fall == 1 ? W.dq3 : nullptr;
    const float *src[8] = {delz, den_, W.denfac, W.t, fall == 0 ? W.workr : fall == 1 ? W.worka : W.vti, dq, W.zi, dqb};
    const int narr = fall == 1 ? 8 : 7;
    for (int a = 0; a < narr; ++a)
        for (int e = threadIdx.x; e < km * W6_TC; e += W6_NT) {
            const int k = e / W6_TC, ci = e % W6_TC;
            if (ci < ncol) w6_lds[a * plane + k * LS + ci] = src[a][d.idx(ib + ci, k0 + k, j)];
        }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane < km ? lane : km - 1;                        // lanes beyond the column read a valid level (values unused)
    const size_t n2 = (size_t)d.nx * d.ny;
    for (int t = 0; t < W6_TC / (W6_NT / 64); ++t) {
        const int ci = wave * (W6_TC / (W6_NT / 64)) + t;
        if (ci >= ncol) break;                                       // wave-uniform
        const float dz = w6_lds[0 * plane + kl * LS + ci], den = w6_lds[1 * plane + kl * LS + ci], denfac = w6_lds[2 * plane + kl * LS + ci],
                    tk = w6_lds[3 * plane + kl * LS + ci], wwl = w6_lds[4 * plane + kl * LS + ci], rql = w6_lds[5 * plane + kl * LS + ci];
        const float rql2 = fall == 1 ? w6_lds[7 * plane + kl * LS + ci] : 0.0f;
        const int kz = (lane <= km ? lane : km) - 1;
        const float zi = lane == 0 ? 0.0f : w6_lds[6 * plane + kz * LS + ci];
        float qn, qn2, pr[2];
        if (fall == 0) w6_fall_wave<0>(C, km, lane, dz, den, denfac, tk, wwl, rql, 0.f, zi, dt, qn, qn2, pr);
        else if (fall == 1) w6_fall_wave<2>(C, km, lane, dz, den, denfac, tk, wwl, rql, rql2, zi, dt, qn, qn2, pr);
        else w6_fall_wave<1>(C, km, lane, dz, den, denfac, tk, wwl, rql, 0.f, zi, dt, qn, qn2, pr);
        if (lane < km) { w6_lds[5 * plane + lane * LS + ci] = qn; if (fall == 1) w6_lds[7 * plane + lane * LS + ci] = qn2; }
        if (lane == 0) {
            const size_t c2 = (size_t)(ib + ci) + (size_t)d.nx * j;
            if (fall == 0) W.delq[c2] = pr[0];
            else if (fall == 1) { W.delq[n2 + c2] = pr[0]; W.delq[2 * n2 + c2] = pr[1]; }
            else W.delq[3 * n2 + c2] = pr[0];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < km * W6_TC; e += W6_NT) {
        const int k = e / W6_TC, ci = e % W6_TC;
        if (ci < ncol) {
            const int c = d.idx(ib + ci, k0 + k, j);
            dq[c] = w6_lds[5 * plane + k * LS + ci];                 // rql(i,:) = qn(:)
            if (fall == 1) dqb[c] = w6_lds[7 * plane + k * LS + ci];
        }
    }
}

// per column: the surface sums of this minor loop from the four fall integrals (:586-588, :666, :672-697)
__global__ void __launch_bounds__(64)
k_w6_surface(Dims d, W6Args A, W6Work W, const float *__restrict__ delz, float dtcld, W6Tiles tl, int k0)
{
    W6_COLUMN_INDEX
    const int c0 = d.idx(i, k0, j), c2 = i + d.nx * j;
    const size_t n2 = (size_t)d.nx * d.ny;
    const float dz0 = delz[c0];
    const float fall1 = W.delq[c2] / dz0 / dtcld, fall2 = W.delq[n2 + c2] / dz0 / dtcld, fall3 = W.delq[2 * n2 + c2] / dz0 / dtcld;
    const float fallc = W.delq[3 * n2 + c2] / dz0 / dtcld;
    const float fallsum = fall1 + fall2 + fall3 + fallc;
    const float fallsum_qsi = fall2 + fallc;
    const float fallsum_qg = fall3;
    if (fallsum > 0.f) W.rain[c2] = fallsum * dz0 / A.denr * dtcld * 1000.f + W.rain[c2];
    if (fallsum_qsi > 0.f) W.snow[c2] = fallsum_qsi * dz0 / A.denr * dtcld * 1000.f + W.snow[c2];
    if (fallsum_qg > 0.f) W.graupel[c2] = fallsum_qg * dz0 / A.denr * dtcld * 1000.f + W.graupel[c2];
}

// per cell, top of a minor loop
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_w6_prep(Dims d, W6Consts C, W6Args A, W6Work W, const float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ q,
          float *__restrict__ qc, float *__restrict__ qi, float *__restrict__ qr, float *__restrict__ qs, float *__restrict__ qg,
          const float *__restrict__ den, const float *__restrict__ p, W6Tiles tl, int k0, int km)
{
    W6_CELL_INDEX
    const int c = d.idx(i, k, j);
    float t, qr_ = qr[c], qs_ = qs[c], qg_ = qg[c], qi_ = qi[c];
    const float q_ = q[c], dn = den[c], pp = p[c];
    if (FIRST) {
        t = th[c] * pii[c];                                                    // :151
        float qc_ = qc[c];
        qc_ = mx(qc_, 0.0f); qr_ = mx(qr_, 0.0f); qi_ = mx(qi_, 0.0f); qs_ = mx(qs_, 0.0f); qg_ = mx(qg_, 0.0f);        // :373-381
        qc[c] = qc_; qr[c] = qr_; qi[c] = qi_; qs[c] = qs_; qg[c] = qg_;
        if (k == k0) { const int c2 = i + d.nx * j; W.rain[c2] = 0.f; W.snow[c2] = 0.f; W.graupel[c2] = 0.f; }   // process_subdomain: precipitation = 0 ...
        W.cpm[c] = A.cpd * (1.f - mx(q_, A.qmin)) + mx(q_, A.qmin) * A.cpv;   // cpmcal :353
        W.xl[c] = A.xlv0 - C.xlv1 * (t - A.t0c);                               // xlcal :354
        W.t[c] = t;
    } else t = W.t[c];
    float tv = 1.f / dn;                                                       // :438-446
    tv = tv * A.den0;
    const float denfac = sqrtf(tv);
    const W6Sat S = w6_sat_coeffs(A);
    const float qs1 = w6_qsat<false>(A, S, t, pp), qs2 = w6_qsat<true>(A, S, t, pp);
    W.denfac[c] = denfac; W.qs1[c] = qs1; W.qs2[c] = qs2;
    W.rh1[c] = mx(q_ / qs1, A.qmin); W.rh2[c] = mx(q_ / qs2, A.qmin);
    W.xni[c] = w6_xni(A, dn, qi_);
    const float vr = w6_slope<0>(C, qr_, dn, denfac, t).vt, vs = w6_slope<1>(C, qs_, dn, denfac, t).vt, vg = w6_slope<2>(C, qg_, dn, denfac, t).vt;
    float workr = vr, worka;                                                   // :557-572
    const float qsum = mx((qs_ + qg_), 1.E-15f);
    if (qsum > 1.e-15f) worka = (vs * qs_ + vg * qg_) / qsum; else worka = 0.f;
    if (qr_ <= 0.0f) workr = 0.0f;
    W.workr[c] = workr; W.worka[c] = worka;
    W.dq1[c] = dn * qr_; W.dq2[c] = dn * qs_; W.dq3[c] = dn * qg_;
}

// per column: blockIdx.y = 0 rain, 1 snow + graupel
__global__ void __launch_bounds__(64)
k_w6_fall(Dims d, W6Consts C, W6Work W, const float *__restrict__ den, const float *__restrict__ delz, float dtcld, W6Tiles tl, int k0, int km)
{
    W6_COLUMN_INDEX
    const int c0 = d.idx(i, k0, j), c2 = i + d.nx * j;
    const size_t n2 = (size_t)d.nx * d.ny;
    float pr[2];
    if (blockIdx.y == 0) {
        w6_fall_column<0>(C, km, d.sk, den + c0, W.denfac + c0, W.t + c0, delz + c0, W.workr + c0, W.dq1 + c0, nullptr, dtcld, pr);
        W.delq[c2] = pr[0];
    } else {
        w6_fall_column<2>(C, km, d.sk, den + c0, W.denfac + c0, W.t + c0, delz + c0, W.worka + c0, W.dq2 + c0, W.dq3 + c0, dtcld, pr);
        W.delq[n2 + c2] = pr[0]; W.delq[2 * n2 + c2] = pr[1];
    }
}

// per cell, between the falls
__global__ void __launch_bounds__(256)
k_w6_melt(Dims d, W6Consts C, W6Args A, W6Work W, const float *__restrict__ qi, float *__restrict__ qr, float *__restrict__ qs, float *__restrict__ qg,
          const float *__restrict__ den, const float *__restrict__ p, float dtcld, W6Tiles tl, int k0, int km)
{
    W6_CELL_INDEX
    const int c = d.idx(i, k, j);
    const float dn = den[c], pp = p[c], denfac = W.denfac[c], cpm = W.cpm[c];
    float t = W.t[c];
    float qr_ = mx(W.dq1[c] / dn, 0.f), qs_ = mx(W.dq2[c] / dn, 0.f), qg_ = mx(W.dq3[c] / dn, 0.f);       // :578-580
    const Slope sr = w6_slope<0>(C, qr_, dn, denfac, t), ss = w6_slope<1>(C, qs_, dn, denfac, t), sg = w6_slope<2>(C, qg_, dn, denfac, t);   // :596
    { float tmp = sr.r3; tmp = tmp * tmp * sr.r; W.frz[c] = tmp; }           // what pgfrz (:747-748) reads of these slopes
    const float supcol = A.t0c - t;
    const float n0sfac = mx(mn(e6(W6_alpha * supcol), W6_n0smax / W6_n0s), 1.f);
    if (t > A.t0c) {                                                          // :603-636
        const float xlf = A.xlf0;
        const float work2 = W6_VENFAC(pp, t, dn);
        if (qs_ > 0.f) {
            const float coeres = ss.r2 * sqrtf(ss.r * ss.rb);
            float psmlt = W6_XKA(t, dn) / xlf * (A.t0c - t) * C.pi / 2.f * n0sfac * (C.precs1 * ss.r2 + C.precs2 * work2 * coeres);
            psmlt = mn(mx(psmlt * dtcld / 1.f, -qs_ / 1.f), 0.f);              // mstep(i) = 1
            qs_ = qs_ + psmlt;
            qr_ = qr_ - psmlt;
            t = t + xlf / cpm * psmlt;
        }
        if (qg_ > 0.f) {
            const float coeres = sg.r2 * sqrtf(sg.r * sg.rb);
            float pgmlt = W6_XKA(t, dn) / xlf * (A.t0c - t) * (C.precg1 * sg.r2 + C.precg2 * work2 * coeres);
            pgmlt = mn(mx(pgmlt * dtcld / 1.f, -qg_ / 1.f), 0.f);
            qg_ = qg_ + pgmlt;
            qr_ = qr_ - pgmlt;
            t = t + xlf / cpm * pgmlt;
        }
    }
    qr[c] = qr_; qs[c] = qs_; qg[c] = qg_; W.t[c] = t;
    const float qi_ = qi[c];                                                  // :641-658
    float vti;
    if (qi_ <= 0.f) vti = 0.f;
    else {
        const float xmi = dn * qi_ / W.xni[c];
        const float diameter = mx(mn(W6_dicon * sqrtf(xmi), W6_dimax), 1.e-25f);
        vti = 1.49e4f * e6(l6(diameter) * (1.31f));
    }
    W.vti[c] = vti; W.dqi[c] = dn * qi_;
}

// per column: the fall of cloud ice and the surface sums of this minor loop (:659-697)
__global__ void __launch_bounds__(64)
k_w6_icefall(Dims d, W6Consts C, W6Args A, W6Work W, const float *__restrict__ den, const float *__restrict__ delz, float dtcld,
             W6Tiles tl, int k0, int km)
{
    W6_COLUMN_INDEX
    const int c0 = d.idx(i, k0, j), c2 = i + d.nx * j;
    const size_t n2 = (size_t)d.nx * d.ny;
    float pr[2];
    w6_fall_column<1>(C, km, d.sk, den + c0, W.denfac + c0, W.t + c0, delz + c0, W.vti + c0, W.dqi + c0, nullptr, dtcld, pr);
    const float dz0 = delz[c0];
    const float fall1 = W.delq[c2] / dz0 / dtcld, fall2 = W.delq[n2 + c2] / dz0 / dtcld, fall3 = W.delq[2 * n2 + c2] / dz0 / dtcld;   // :586-588
    const float fallc = pr[0] / dz0 / dtcld;                                  // :666
    const float fallsum = fall1 + fall2 + fall3 + fallc;
    const float fallsum_qsi = fall2 + fallc;
    const float fallsum_qg = fall3;
    if (fallsum > 0.f) W.rain[c2] = fallsum * dz0 / A.denr * dtcld * 1000.f + W.rain[c2];
    if (fallsum_qsi > 0.f) W.snow[c2] = fallsum_qsi * dz0 / A.denr * dtcld * 1000.f + W.snow[c2];
    if (fallsum_qg > 0.f) W.graupel[c2] = fallsum_qg * dz0 / A.denr * dtcld * 1000.f + W.graupel[c2];
}

// per cell: everything after the falls; LAST: th = t / pii (:171)
template <bool LAST>
__global__ void __launch_bounds__(256)
k_w6_rates(Dims d, W6Consts C, W6Args A, W6Work W, float *__restrict__ th, const float *__restrict__ pii, float *__restrict__ q,
           float *__restrict__ qc, float *__restrict__ qi, float *__restrict__ qr, float *__restrict__ qs, float *__restrict__ qg,
           const float *__restrict__ den, const float *__restrict__ p, float dtcld, W6Tiles tl, int k0, int km)
{
    W6_CELL_INDEX
    const int c = d.idx(i, k, j);
    const float dn = den[c], pp = p[c], denfac = W.denfac[c], cpm = W.cpm[c], xl = W.xl[c], qs1 = W.qs1[c], qs2 = W.qs2[c], rh1 = W.rh1[c], rh2 = W.rh2[c];
    const float t0c = A.t0c, qmin = A.qmin, xls = A.xls, pi = C.pi, denr = A.denr;
    float t = W.t[c], qv = q[c], qc_ = qc[c], qr_ = qr[c], qs_ = qs[c], qg_ = qg[c];
    float qi_ = mx(W.dqi[c] / dn, 0.f);                                        // :661
    {   // instant melting / freezing :703-759
        const float supcol = t0c - t;
        float xlf = xls - xl;
        if (supcol < 0.f) xlf = A.xlf0;
        if (supcol < 0.f && qi_ > 0.f) { qc_ = qc_ + qi_; t = t - xlf / cpm * qi_; qi_ = 0.f; }
        if (supcol > 40.f && qc_ > 0.f) { qi_ = qi_ + qc_; t = t + xlf / cpm * qc_; qc_ = 0.f; }
        if (supcol > 0.f && qc_ > qmin) {
            const float supcolt = mn(supcol, 50.f);
            const float pfrzdtc = mn(W6_pfrz1 * (e6(W6_pfrz2 * supcolt) - 1.f) * dn / denr / W6_xncr * qc_ * qc_ * dtcld, qc_);
            qi_ = qi_ + pfrzdtc; t = t + xlf / cpm * pfrzdtc; qc_ = qc_ - pfrzdtc;
        }
        if (supcol > 0.f && qr_ > 0.f) {
            const float temp = W.frz[c];
            const float supcolt = mn(supcol, 50.f);
            const float pfrzdtr = mn(20.f * (pi * pi) * W6_pfrz1 * W6_n0r * denr / dn * (e6(W6_pfrz2 * supcolt) - 1.f) * temp * dtcld, qr_);
            qg_ = qg_ + pfrzdtr; t = t + xlf / cpm * pfrzdtr; qr_ = qr_ - pfrzdtr;
        }
    }
    const Slope R = w6_slope<0>(C, qr_, dn, denfac, t), S = w6_slope<1>(C, qs_, dn, denfac, t), G = w6_slope<2>(C, qg_, dn, denfac, t);   // :765-773
    const float work1a = W6_DIFFAC(xl, pp, t, dn, qs1), work1b = W6_DIFFAC(xls, pp, t, dn, qs2), work2 = W6_VENFAC(pp, t, dn);           // :782-789
    float prevp = 0.f, psdep = 0.f, pgdep = 0.f, praut = 0.f, psaut = 0.f, pgaut = 0.f, pracw = 0.f, praci = 0.f, piacr = 0.f, psaci = 0.f,
          psacw = 0.f, pracs = 0.f, psacr = 0.f, pgacw = 0.f, paacw = 0.f, pgaci = 0.f, pgacr = 0.f, pgacs = 0.f, pigen = 0.f, pidep = 0.f,
          pseml = 0.f, pgeml = 0.f, psevp = 0.f, pgevp = 0.f;
    {   // warm rain :802-840
        const float supsat = mx(qv, qmin) - qs1;
        const float satdt = supsat / dtcld;
        if (qc_ > C.qc0) { praut = C.qck1 * p6(qc_, 7.f / 3.f); praut = mn(praut, qc_ / dtcld); }
        if (qr_ > W6_qcrmin && qc_ > qmin) pracw = mn(C.pacrr * R.r3 * R.rb * qc_ * denfac, qc_ / dtcld);
        if (qr_ > 0.f) {
            const float coeres = R.r2 * sqrtf(R.r * R.rb);
            prevp = (rh1 - 1.f) * (C.precr1 * R.r2 + C.precr2 * work2 * coeres) / work1a;
            if (prevp < 0.f) { prevp = mx(prevp, -qr_ / dtcld); prevp = mx(prevp, satdt / 2); }
            else prevp = mn(prevp, satdt / 2);
        }
    }
    // cold rain :855-1128
    const float supcol = t0c - t;
    const float n0sfac = mx(mn(e6(W6_alpha * supcol), W6_n0smax / W6_n0s), 1.f);
    const float supsat = mx(qv, qmin) - qs2;
    const float satdt = supsat / dtcld;
    int ifsat = 0;
    const float xni = w6_xni(A, dn, qi_);
    const float eacrs = e6(0.07f * (-supcol));
    const float xmi = dn * qi_ / xni;
    const float diameter = mn(W6_dicon * sqrtf(xmi), W6_dimax);
    const float vt2i = 1.49e4f * p6(diameter, 1.31f);
    const float vt2r = C.pvtr * R.rb * denfac, vt2s = C.pvts * S.rb * denfac, vt2g = C.pvtg * G.rb * denfac;
    const float qsum = mx((qs_ + qg_), 1.E-15f);
    float vt2ave;
    if (qsum > 1.e-15f) vt2ave = (vt2s * qs_ + vt2g * qg_) / (qsum); else vt2ave = 0.f;
    if (supcol > 0.f && qi_ > qmin) {
        if (qr_ > W6_qcrmin) {
            const float acrfac = 2.f * R.r3 + 2.f * diameter * R.r2 + diameter * diameter * R.r;
            praci = pi * qi_ * W6_n0r * fabsf(vt2r - vt2i) * acrfac / 4.f;
            praci = mn(praci, qi_ / dtcld);
            piacr = pi * pi * W6_avtr * W6_n0r * denr * xni * denfac * C.g6pbr * R.r3 * R.r3 * R.rb / 24.f / dn;
            piacr = mn(piacr, qr_ / dtcld);
        }
        if (qs_ > W6_qcrmin) {
            const float acrfac = 2.f * S.r3 + 2.f * diameter * S.r2 + diameter * diameter * S.r;
            psaci = pi * qi_ * eacrs * W6_n0s * n0sfac * fabsf(vt2ave - vt2i) * acrfac / 4.f;
            psaci = mn(psaci, qi_ / dtcld);
        }
        if (qg_ > W6_qcrmin) {
            const float egi = e6(0.07f * (-supcol));
            const float acrfac = 2.f * G.r3 + 2.f * diameter * G.r2 + diameter * diameter * G.r;
            pgaci = pi * egi * qi_ * W6_n0g * fabsf(vt2ave - vt2i) * acrfac / 4.f;
            pgaci = mn(pgaci, qi_ / dtcld);
        }
    }
    if (qs_ > W6_qcrmin && qc_ > qmin) psacw = mn(C.pacrc * n0sfac * S.r3 * S.rb * qc_ * denfac, qc_ / dtcld);
    if (qg_ > W6_qcrmin && qc_ > qmin) pgacw = mn(C.pacrg * G.r3 * G.rb * qc_ * denfac, qc_ / dtcld);
    if (qsum > 1.e-15f) paacw = (qs_ * psacw + qg_ * pgacw) / (qsum);
    if (qs_ > W6_qcrmin && qr_ > W6_qcrmin) {
        if (supcol > 0) {
            const float acrfac = 5.f * S.r3 * S.r3 * R.r + 2.f * S.r3 * S.r2 * R.r2 + .5f * S.r2 * S.r2 * R.r3;
            pracs = pi * pi * W6_n0r * W6_n0s * n0sfac * fabsf(vt2r - vt2ave) * (W6_dens / dn) * acrfac;
            pracs = mn(pracs, qs_ / dtcld);
        }
        const float acrfac = 5.f * R.r3 * R.r3 * S.r + 2.f * R.r3 * R.r2 * S.r2 + .5f * R.r2 * R.r2 * S.r3;
        psacr = pi * pi * W6_n0r * W6_n0s * n0sfac * fabsf(vt2ave - vt2r) * (denr / dn) * acrfac;
        psacr = mn(psacr, qr_ / dtcld);
    }
    if (qg_ > W6_qcrmin && qr_ > W6_qcrmin) {
        const float acrfac = 5.f * R.r3 * R.r3 * G.r + 2.f * R.r3 * R.r2 * G.r2 + .5f * R.r2 * R.r2 * G.r3;
        pgacr = pi * pi * W6_n0r * W6_n0g * fabsf(vt2ave - vt2r) * (denr / dn) * acrfac;
        pgacr = mn(pgacr, qr_ / dtcld);
    }
    if (qg_ > W6_qcrmin && qs_ > W6_qcrmin) pgacs = 0.f;
    if (supcol <= 0) {
        const float xlf = A.xlf0;
        if (qs_ > 0.f) pseml = mn(mx(A.cliq * supcol * (paacw + psacr) / xlf, -qs_ / dtcld), 0.f);
        if (qg_ > 0.f) pgeml = mn(mx(A.cliq * supcol * (paacw + pgacr) / xlf, -qg_ / dtcld), 0.f);
    }
    if (supcol > 0) {
        if (qi_ > 0 && ifsat != 1) {
            pidep = 4.f * diameter * xni * (rh2 - 1.f) / work1b;
            const float supice = satdt - prevp;
            if (pidep < 0.f) { pidep = mx(mx(pidep, satdt / 2), supice); pidep = mx(pidep, -qi_ / dtcld); }
            else pidep = mn(mn(pidep, satdt / 2), supice);
            if (fabsf(prevp + pidep) >= fabsf(satdt)) ifsat = 1;
        }
        if (qs_ > 0.f && ifsat != 1) {
            const float coeres = S.r2 * sqrtf(S.r * S.rb);
            psdep = (rh2 - 1.f) * n0sfac * (C.precs1 * S.r2 + C.precs2 * work2 * coeres) / work1b;
            const float supice = satdt - prevp - pidep;
            if (psdep < 0.f) { psdep = mx(psdep, -qs_ / dtcld); psdep = mx(mx(psdep, satdt / 2), supice); }
            else psdep = mn(mn(psdep, satdt / 2), supice);
            if (fabsf(prevp + pidep + psdep) >= fabsf(satdt)) ifsat = 1;
        }
        if (qg_ > 0.f && ifsat != 1) {
            const float coeres = G.r2 * sqrtf(G.r * G.rb);
            pgdep = (rh2 - 1.f) * (C.precg1 * G.r2 + C.precg2 * work2 * coeres) / work1b;
            const float supice = satdt - prevp - pidep - psdep;
            if (pgdep < 0.f) { pgdep = mx(pgdep, -qg_ / dtcld); pgdep = mx(mx(pgdep, satdt / 2), supice); }
            else pgdep = mn(mn(pgdep, satdt / 2), supice);
            if (fabsf(prevp + pidep + psdep + pgdep) >= fabsf(satdt)) ifsat = 1;
        }
        if (supsat > 0 && ifsat != 1) {
            const float supice = satdt - prevp - pidep - psdep - pgdep;
            const float xni0 = 1.e3f * e6(0.1f * supcol);
            const float roqi0 = 4.92e-11f * p6(xni0, 1.33f);
            pigen = mx(0.f, (roqi0 / dn - mx(qi_, 0.f)) / dtcld);
            pigen = mn(mn(pigen, satdt), supice);
        }
        if (qi_ > 0.f) { const float qimax = C.roqimax / dn; psaut = mx(0.f, (qi_ - qimax) / dtcld); }
        if (qs_ > 0.f) {
            const float alpha2 = 1.e-3f * e6(0.09f * (-supcol));
            pgaut = mn(mx(0.f, alpha2 * (qs_ - W6_qs0)), qs_ / dtcld);
        }
    }
    if (supcol < 0.f) {
        if (qs_ > 0.f && rh1 < 1.f) {
            const float coeres = S.r2 * sqrtf(S.r * S.rb);
            psevp = (rh1 - 1.f) * n0sfac * (C.precs1 * S.r2 + C.precs2 * work2 * coeres) / work1a;
            psevp = mn(mx(psevp, -qs_ / dtcld), 0.f);
        }
        if (qg_ > 0.f && rh1 < 1.f) {
            const float coeres = G.r2 * sqrtf(G.r * G.rb);
            pgevp = (rh1 - 1.f) * (C.precg1 * G.r2 + C.precg2 * work2 * coeres) / work1a;
            pgevp = mn(mx(pgevp, -qg_ / dtcld), 0.f);
        }
    }
    // conservation of the source terms and feedback :1136-1318
    float delta2 = 0.f, delta3 = 0.f, value, source, factor;
    if (qr_ < 1.e-4f && qs_ < 1.e-4f) delta2 = 1.f;
    if (qr_ < 1.e-4f) delta3 = 1.f;
    if (t <= t0c) {
        value = mx(qmin, qc_);
        source = (praut + pracw + paacw + paacw) * dtcld;
        if (source > value) { factor = value / source; praut = praut * factor; pracw = pracw * factor; paacw = paacw * factor; }
        value = mx(qmin, qi_);
        source = (psaut - pigen - pidep + praci + psaci + pgaci) * dtcld;
        if (source > value) {
            factor = value / source;
            psaut = psaut * factor; pigen = pigen * factor; pidep = pidep * factor; praci = praci * factor; psaci = psaci * factor; pgaci = pgaci * factor;
        }
        value = mx(qmin, qr_);
        source = (-praut - prevp - pracw + piacr + psacr + pgacr) * dtcld;
        if (source > value) {
            factor = value / source;
            praut = praut * factor; prevp = prevp * factor; pracw = pracw * factor; piacr = piacr * factor; psacr = psacr * factor; pgacr = pgacr * factor;
        }
        value = mx(qmin, qs_);
        source = -(psdep + psaut - pgaut + paacw + piacr * delta3 + praci * delta3 - pracs * (1.f - delta2) + psacr * delta2 + psaci - pgacs) * dtcld;
        if (source > value) {
            factor = value / source;
            psdep = psdep * factor; psaut = psaut * factor; pgaut = pgaut * factor; paacw = paacw * factor; piacr = piacr * factor;
            praci = praci * factor; psaci = psaci * factor; pracs = pracs * factor; psacr = psacr * factor; pgacs = pgacs * factor;
        }
        value = mx(qmin, qg_);
        source = -(pgdep + pgaut + piacr * (1.f - delta3) + praci * (1.f - delta3) + psacr * (1.f - delta2) + pracs * (1.f - delta2)
                   + pgaci + paacw + pgacr + pgacs) * dtcld;
        if (source > value) {
            factor = value / source;
            pgdep = pgdep * factor; pgaut = pgaut * factor; piacr = piacr * factor; praci = praci * factor; psacr = psacr * factor;
            pracs = pracs * factor; paacw = paacw * factor; pgaci = pgaci * factor; pgacr = pgacr * factor; pgacs = pgacs * factor;
        }
        const float w2 = -(prevp + psdep + pgdep + pigen + pidep);
        qv = qv + w2 * dtcld;
        qc_ = mx(qc_ - (praut + pracw + paacw + paacw) * dtcld, 0.f);
        const float qr_new = mx(qr_ + (praut + pracw + prevp - piacr - pgacr - psacr) * dtcld, 0.f);
        qi_ = mx(qi_ - (psaut + praci + psaci + pgaci - pigen - pidep) * dtcld, 0.f);
        const float qs_new = mx(qs_ + (psdep + psaut + paacw - pgaut + piacr * delta3 + praci * delta3 + psaci - pgacs - pracs * (1.f - delta2)
                                       + psacr * delta2) * dtcld, 0.f);
        const float qg_new = mx(qg_ + (pgdep + pgaut + piacr * (1.f - delta3) + praci * (1.f - delta3) + psacr * (1.f - delta2)
                                       + pracs * (1.f - delta2) + pgaci + paacw + pgacr + pgacs) * dtcld, 0.f);
        qr_ = qr_new; qs_ = qs_new; qg_ = qg_new;
        const float xlf = xls - xl;
        const float xlwork2 = -xls * (psdep + pgdep + pidep + pigen) - xl * prevp - xlf * (piacr + paacw + paacw + pgacr + psacr);
        t = t - xlwork2 / cpm * dtcld;
    } else {
        value = mx(qmin, qc_);
        source = (praut + pracw + paacw + paacw) * dtcld;
        if (source > value) { factor = value / source; praut = praut * factor; pracw = pracw * factor; paacw = paacw * factor; }
        value = mx(qmin, qr_);
        source = (-paacw - praut + pseml + pgeml - pracw - paacw - prevp) * dtcld;
        if (source > value) {
            factor = value / source;
            praut = praut * factor; prevp = prevp * factor; pracw = pracw * factor; paacw = paacw * factor; pseml = pseml * factor; pgeml = pgeml * factor;
        }
        value = mx(W6_qcrmin, qs_);
        source = (pgacs - pseml - psevp) * dtcld;
        if (source > value) { factor = value / source; pgacs = pgacs * factor; psevp = psevp * factor; pseml = pseml * factor; }
        value = mx(W6_qcrmin, qg_);
        source = -(pgacs + pgevp + pgeml) * dtcld;
        if (source > value) { factor = value / source; pgacs = pgacs * factor; pgevp = pgevp * factor; pgeml = pgeml * factor; }
        const float w2 = -(prevp + psevp + pgevp);
        qv = qv + w2 * dtcld;
        qc_ = mx(qc_ - (praut + pracw + paacw + paacw) * dtcld, 0.f);
        qr_ = mx(qr_ + (praut + pracw + prevp + paacw + paacw - pseml - pgeml) * dtcld, 0.f);
        qs_ = mx(qs_ + (psevp - pgacs + pseml) * dtcld, 0.f);
        qg_ = mx(qg_ + (pgacs + pgevp + pgeml) * dtcld, 0.f);
        const float xlf = xls - xl;
        const float xlwork2 = -xl * (prevp + psevp + pgevp) - xlf * (pseml + pgeml);
        t = t - xlwork2 / cpm * dtcld;
    }
    {   // saturation adjustment :1330-1385
        const W6Sat Sc = w6_sat_coeffs(A);
        const float qsw = w6_qsat<false>(A, Sc, t, pp);
        const float w1 = (mx(qv, qmin) - qsw) / (1.f + xl * xl / (A.rv * cpm) * qsw / (t * t));      // conden :365
        float pcond = mn(mx(w1 / dtcld, 0.f), mx(qv, 0.f) / dtcld);
        if (qc_ > 0.f && w1 < 0.f) pcond = mx(w1, -qc_) / dtcld;
        qv = qv - pcond * dtcld;
        qc_ = mx(qc_ + pcond * dtcld, 0.f);
        t = t + pcond * xl / cpm * dtcld;
        if (qc_ <= qmin) qc_ = 0.0f;
        if (qi_ <= qmin) qi_ = 0.0f;
    }
    q[c] = qv; qc[c] = qc_; qi[c] = qi_; qr[c] = qr_; qs[c] = qs_; qg[c] = qg_;
    if (LAST) th[c] = t / pii[c]; else W.t[c] = t;
}

// mp_driver.f90:587-595: REAL(8) accumulators += this call's REAL(4) precipitation / snowfall / graupel
__global__ void k_w6_accumulate(Dims d, W6Work W, double *__restrict__ precip_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                                W6Tiles tl)
{
    W6_COLUMN_INDEX
    const int c2 = i + d.nx * j;
    precip_acc[c2] = precip_acc[c2] + W.rain[c2];
    snow_acc[c2] = snow_acc[c2] + W.snow[c2];
    graupel_acc[c2] = graupel_acc[c2] + W.graupel[c2];
}

// rgmma (:1386-1405): the 10000-term product form of 1/Gamma, host libm like the compiled reference
float w6_rgmma(float x)
{
    const float euler = 0.577215664901532f;
    if (x == 1.f) return 0.f;
    float r = x * expf(euler * x);
    for (int i = 1; i <= 10000; ++i) { const float y = (float)i; r = r * (1.000f + x / y) * expf(-x / y); }
    return 1.f / r;
}
}  // namespace

void icar_wsm6_free(icar_hip_ctx *c)
{
    if (!c->wsm6) return;
    W6Work &w = c->wsm6->w;
    float *ps[] = {w.t, w.cpm, w.xl, w.denfac, w.qs1, w.qs2, w.rh1, w.rh2, w.xni, w.workr, w.worka, w.dq1, w.dq2, w.dq3, w.vti, w.dqi, w.frz, w.zi,
                   w.rain, w.snow, w.graupel, w.delq};
    for (float *p : ps) if (p) hipFree(p);
    delete c->wsm6; c->wsm6 = nullptr;
}

int icar_wsm6_init_run(icar_hip_ctx *c)
{
    // wsm6init(rhoair0, rhowater, rhosnow, cliq, cpv) as mp_driver.f90:100 calls it (wrf_constants.f90:30-35, :65-67), REAL(4)
    // arithmetic in the reference's order (:1432-1506)
    if (!c->wsm6) c->wsm6 = new Wsm6State;
    W6Consts &K = c->wsm6->c;
    const float den0 = 1.28f, denr = 1000.f, dens = 100.f, cl = 4190.f, cpv = 4.f * 461.6f;
    K.pi = 4.f * atanf(1.f);
    K.xlv1 = cl - cpv;
    K.qc0 = 4.f / 3.f * K.pi * denr * (W6_r0 * W6_r0 * W6_r0) * W6_xncr / den0;
    K.qck1 = .104f * 9.8f * W6_peaut / powf(W6_xncr * denr, 1.f / 3.f) / W6_xmyu * powf(den0, 4.f / 3.f);
    const float bvtr2 = 2.5f + .5f * W6_bvtr, bvtr3 = 3.f + W6_bvtr, bvtr4 = 4.f + W6_bvtr, bvtr6 = 6.f + W6_bvtr;
    const float g3pbr = w6_rgmma(bvtr3), g4pbr = w6_rgmma(bvtr4), g5pbro2 = w6_rgmma(bvtr2);
    K.g6pbr = w6_rgmma(bvtr6);
    K.pvtr = W6_avtr * g4pbr / 6.f;
    const float eacrr = 1.0f;
    K.pacrr = K.pi * W6_n0r * W6_avtr * g3pbr * .25f * eacrr;
    K.precr1 = 2.f * K.pi * W6_n0r * .78f;
    K.precr2 = 2.f * K.pi * W6_n0r * .31f * powf(W6_avtr, .5f) * g5pbro2;
    { const float d2 = W6_dimax * W6_dimax, d4 = d2 * d2; K.roqimax = 2.08e22f * (d4 * d4); }
    const float bvts2 = 2.5f + .5f * W6_bvts, bvts3 = 3.f + W6_bvts, bvts4 = 4.f + W6_bvts;
    const float g3pbs = w6_rgmma(bvts3), g4pbs = w6_rgmma(bvts4), g5pbso2 = w6_rgmma(bvts2);
    K.pvts = W6_avts * g4pbs / 6.f;
    K.precs1 = 4.f * W6_n0s * .65f;
    K.precs2 = 4.f * W6_n0s * .44f * powf(W6_avts, .5f) * g5pbso2;
    K.pidn0r = K.pi * denr * W6_n0r;
    K.pidn0s = K.pi * dens * W6_n0s;
    K.pacrc = K.pi * W6_n0s * W6_avts * g3pbs * .25f * W6_eacrc;
    const float bvtg2 = 2.5f + .5f * W6_bvtg, bvtg3 = 3.f + W6_bvtg, bvtg4 = 4.f + W6_bvtg;
    const float g3pbg = w6_rgmma(bvtg3), g4pbg = w6_rgmma(bvtg4);
    K.pacrg = K.pi * W6_n0g * W6_avtg * g3pbg * .25f;
    const float g5pbgo2 = w6_rgmma(bvtg2);
    K.pvtg = W6_avtg * g4pbg / 6.f;
    K.precg1 = 2.f * K.pi * W6_n0g * .78f;
    K.precg2 = 2.f * K.pi * W6_n0g * .31f * powf(W6_avtg, .5f) * g5pbgo2;
    K.pidn0g = K.pi * W6_deng * W6_n0g;
    const float lam[3] = {W6_lamdarmax, W6_lamdasmax, W6_lamdagmax}, bv[3] = {W6_bvtr, W6_bvts, W6_bvtg};
    for (int s = 0; s < 3; ++s) {
        K.smax[s] = 1.f / lam[s];
        K.sbmax[s] = powf(K.smax[s], bv[s]);
        K.s2max[s] = K.smax[s] * K.smax[s];
        K.s3max[s] = K.s2max[s] * K.smax[s];
    }
    c->wsm6->ready = true;
    return 0;
}

int icar_wsm6_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte)
{
    const int tile[1][4] = {{its, ite, jts, jte}};
    return icar_wsm6_run_tiles(c, dt, 1, tile, kts, kte);
}

// up to 4 non-overlapping tiles {its, ite, jts, jte} in ONE sequence of launches (process_halo's strips, mp_driver.f90:609-658)
int icar_wsm6_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte)
{
    if (!c->wsm6 || !c->wsm6->ready) { icar_set_error("wsm6: call icar_hip_wsm6_init first"); return 1; }
    if (ntiles < 0 || ntiles > 4) { icar_set_error("wsm6: 0..4 tiles per call"); return 1; }
    if (kts < c->kms || kte > c->kme) { icar_set_error("wsm6: tile outside memory bounds"); return 1; }
    W6Tiles tl; tl.n = 0; tl.coff[0] = 0; tl.boff[0] = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int its = tiles[t][0], ite = tiles[t][1], jts = tiles[t][2], jte = tiles[t][3];
        if (its < c->ims || ite > c->ime || jts < c->jms || jte > c->jme) { icar_set_error("wsm6: tile outside memory bounds"); return 1; }
        if (ite < its || jte < jts) continue;
        const int n = tl.n, i0 = its - c->ims, i1 = ite - c->ims, j0 = jts - c->jms, nrow = jte - jts + 1;
        for (int o = 0; o < n; ++o)                                   // columns are updated in place: two tiles on one column would race
            if (i0 <= tl.i1[o] && i1 >= tl.i0[o] && j0 < tl.j0[o] + tl.nrow[o] && j0 + nrow > tl.j0[o]) { icar_set_error("wsm6: tiles of one call must not overlap"); return 1; }
        tl.i0[n] = i0; tl.i1[n] = i1; tl.j0[n] = j0; tl.nrow[n] = nrow;
        tl.coff[n + 1] = tl.coff[n] + (i1 - i0 + 1) * nrow;
        tl.boff[n + 1] = tl.boff[n] + ((i1 - i0 + W6_TC) / W6_TC) * nrow;
        ++tl.n;
    }
    if (tl.n == 0) return 0;
    for (int t = tl.n; t < 4; ++t) { tl.i0[t] = tl.i1[t] = tl.j0[t] = 0; tl.nrow[t] = 0; tl.coff[t + 1] = tl.coff[tl.n]; tl.boff[t + 1] = tl.boff[tl.n]; }
    const int km = kte - kts + 1;
    if (km < 4 || km > W6_MAXK) { icar_set_error("wsm6: 4..64 levels in this build"); return 1; }
    float *th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE), *q = icar_field_f(c, ICAR_F_WATER_VAPOR);
    float *qc = icar_field_f(c, ICAR_F_CLOUD_WATER), *qr = icar_field_f(c, ICAR_F_RAIN), *qi = icar_field_f(c, ICAR_F_CLOUD_ICE);
    float *qs = icar_field_f(c, ICAR_F_SNOW), *qg = icar_field_f(c, ICAR_F_GRAUPEL);
    const float *den = icar_field_f(c, ICAR_F_DENSITY), *pii = icar_field_f(c, ICAR_F_EXNER), *p = icar_field_f(c, ICAR_F_PRESSURE), *dz = icar_field_f(c, ICAR_F_DZ_MASS);
    double *pa = (double *)icar_field_f(c, ICAR_F_PRECIPITATION, false), *sa = (double *)icar_field_f(c, ICAR_F_SNOWFALL, false);
    double *ga = (double *)icar_field_f(c, ICAR_F_GRAUPEL_ACC, false);
    if (!th || !q || !qc || !qr || !qi || !qs || !qg || !den || !pii || !p || !dz || !pa || !sa || !ga) return 1;
    Wsm6State *S = c->wsm6;
    W6Work &W = S->w;
    const size_t n2 = (size_t)c->d.nx * c->d.ny;
    if (!W.t) {
        float **p3[] = {&W.t, &W.cpm, &W.xl, &W.denfac, &W.qs1, &W.qs2, &W.rh1, &W.rh2, &W.xni, &W.workr, &W.worka, &W.dq1, &W.dq2, &W.dq3, &W.vti, &W.dqi, &W.frz, &W.zi};
        for (float **x : p3) HIPCHK(hipMalloc(x, c->n3 * sizeof(float)));
        HIPCHK(hipMalloc(&W.rain, n2 * sizeof(float))); HIPCHK(hipMalloc(&W.snow, n2 * sizeof(float))); HIPCHK(hipMalloc(&W.graupel, n2 * sizeof(float)));
        HIPCHK(hipMalloc(&W.delq, 4 * n2 * sizeof(float)));
    }
    // what mp_driver.f90:518-550 passes: gravity, cp, cpv, Rd, Rw, 273.15, EP1, EP2, epsilon, XLS, XLV, XLF, rhoair0, rhowater,
    // cliq, cice, psat (icar_constants.f90:391-420, wrf_constants.f90:10-67)
    W6Args A;
    A.delt = dt; A.g = 9.81f; A.cpd = 1012.0f; A.cpv = 4.f * 461.6f; A.rd = 287.058f; A.rv = 461.5f; A.t0c = 273.15f;
    A.ep1 = 461.5f / 287.058f - 1.f; A.ep2 = 287.058f / 461.5f; A.qmin = 1.e-15f; A.xls = 2.85e6f; A.xlv0 = 2.5e6f; A.xlf0 = 3.50e5f;
    A.den0 = 1.28f; A.denr = 1000.f; A.cliq = 4190.f; A.cice = 2106.f; A.psat = 610.78f;
    // minor time steps :416-418
    const long lp = lroundf(A.delt / W6_dtcldcr);
    const int loops = lp > 1 ? (int)lp : 1;
    float dtcld = A.delt / (float)loops;
    if (A.delt <= W6_dtcldcr) dtcld = A.delt;
    ScopedTimer tm(c, "mp");
    // (the call's REAL(4) surface sums are zeroed per column by k_w6_prep: calls on disjoint tiles -- the strips and the interior on
    // the context's two streams -- share no scratch)
    const int k0 = kts - c->kms;
    const long long ncol = tl.coff[tl.n], ncell = ncol * km;
    const dim3 gc((unsigned)((ncell + 255) / 256)), bc(256), g2((unsigned)((ncol + 63) / 64)), b2(64);
    const bool wave_falls = km + 1 <= 64;                       // lane = level; taller columns: one thread per column
    const dim3 gt(tl.boff[tl.n], 1, 1);
    const size_t tile_lds = 8 * (size_t)km * (W6_TC + 1) * sizeof(float);
    if (wave_falls && tile_lds > 64 * 1024)                     // more than 61 levels: above HIP's default dynamic-LDS limit (160 kB per CU on gfx950)
        HIPCHK(hipFuncSetAttribute((const void *)k_w6_fall_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
    if (wave_falls) hipLaunchKernelGGL(k_w6_zi, g2, b2, 0, c->stream, c->d, dz, W.zi, tl, k0, km);
    for (int loop = 1; loop <= loops; ++loop) {
        if (loop == 1) hipLaunchKernelGGL((k_w6_prep<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qc, qi, qr, qs, qg, den, p, tl, k0, km);
        else           hipLaunchKernelGGL((k_w6_prep<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qc, qi, qr, qs, qg, den, p, tl, k0, km);
        if (wave_falls) hipLaunchKernelGGL(k_w6_fall_tile, dim3(gt.x, 1, 2), dim3(W6_NT), tile_lds, c->stream, c->d, S->c, W, den, dz, dtcld, tl, k0, km, 0);
        else hipLaunchKernelGGL(k_w6_fall, dim3(g2.x, 2), b2, 0, c->stream, c->d, S->c, W, den, dz, dtcld, tl, k0, km);
        hipLaunchKernelGGL(k_w6_melt, gc, bc, 0, c->stream, c->d, S->c, A, W, qi, qr, qs, qg, den, p, dtcld, tl, k0, km);
        if (wave_falls) {
            hipLaunchKernelGGL(k_w6_fall_tile, gt, dim3(W6_NT), tile_lds, c->stream, c->d, S->c, W, den, dz, dtcld, tl, k0, km, 2);
            hipLaunchKernelGGL(k_w6_surface, g2, b2, 0, c->stream, c->d, A, W, dz, dtcld, tl, k0);
        } else hipLaunchKernelGGL(k_w6_icefall, g2, b2, 0, c->stream, c->d, S->c, A, W, den, dz, dtcld, tl, k0, km);
        if (loop == loops) hipLaunchKernelGGL((k_w6_rates<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qc, qi, qr, qs, qg, den, p, dtcld, tl, k0, km);
        else               hipLaunchKernelGGL((k_w6_rates<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qc, qi, qr, qs, qg, den, p, dtcld, tl, k0, km);
    }
    hipLaunchKernelGGL(k_w6_accumulate, g2, b2, 0, c->stream, c->d, W, pa, sa, ga, tl);
    HIPCHK(hipGetLastError());
    return 0;
}
