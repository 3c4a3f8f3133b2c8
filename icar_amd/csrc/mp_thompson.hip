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
#include "column_comm.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

const ThState *icar_thompson_device_state(icar_hip_ctx *c);
const ThState *icar_thompson_host_state(icar_hip_ctx *c);

namespace {
// x**y for the positive bases the scheme uses: exp(y*log(x)) in FP64 (relative error ~1e-14, i.e. the
// float result is the correctly rounded one with probability 1 - 1e-7).
__device__ __forceinline__ double d_pow(double x, double y)
{
    if (y == 0.0) return 1.0;
    if (x > 0.0) return d_exp(y * d_log(x));
    // not reached by the scheme's positive bases; kept out of line of the hot code (ocml's pow is ~230 instructions per
    // call site): 0**y = 0 / +inf like libm, a negative base gives NaN (Fortran: invalid for a REAL exponent)
    if (x == 0.0) return y > 0.0 ? 0.0 : __builtin_inf();
    return __builtin_nan("");
}
__device__ __forceinline__ float d_powf(float x, float y) { return (float)d_pow((double)x, (double)y); }
// 10.**y (REAL y): exp(y ln 10), the same evaluation d_pow makes with its log already folded
__device__ __forceinline__ float d_pow10f(float y) { return y == 0.0f ? 1.0f : (float)d_exp((double)y * 2.30258509299404568402e+00); }
__device__ __forceinline__ float d_expf(float x) { return (float)d_exp((double)x); }
__device__ __forceinline__ float d_log10f(float x)
{
    if (x > 0.0f) return (float)(d_log((double)x) * 4.34294481903251816668e-01);   // log(x) / ln 10
    return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
}

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
__device__ __forceinline__ int dec_index_f(float r, int n2)
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
__device__ __forceinline__ int dec_index_d(double r, int n2)
{
    const int nic = (int)lround(log10(r));
    int n = nic - 1;
    for (int nn = nic - 1; nn <= nic + 1; ++nn) {
        n = nn;
        if ((r / (double)powi10f(nn)) >= 1.0 && (r / (double)powi10f(nn)) < 10.0) break;
    }
    return (int)(r / (double)powi10f(n)) + 10 * (n - n2) - (n - n2);
}

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


template <int KMAX>
__device__ void th_column(const ThState *__restrict__ T, float *qv1d, float *qc1d, float *qi1d, float *qr1d, float *qs1d, float *qg1d, float *ni1d, float *nr1d,
               float *t1d, float *p1d, const float *dzq, float *pptrain, float *pptsnow, float *pptgraul, float *pptice,
               int nz, float dt)
{
    const float *sa = T->sa, *sb = T->sb;
    const float R1 = TH_R1, R2 = TH_R2, eps = TH_eps, T_0 = TH_T_0, PI2 = TH_PI2;
    const float am_r = TH_am_r, am_i = TH_am_i, bm_i = TH_bm_i, bm_g = TH_bm_g, mu_i = TH_mu_i, mu_g = TH_mu_g;
    const float D0r = TH_D0r, D0c = TH_D0c, D0s = TH_D0s, D0g = TH_D0g, fv_r = TH_fv_r, lsub = TH_lsub, lvap0 = TH_lvap0;
    const float oRv = TH_oRv, olfus = TH_olfus, xm0i = TH_xm0i, C_cube = TH_C_cube, HGFR = TH_HGFR, rho_w = TH_rho_w;
    const float mu_r = T->mu_r, mu_c = T->mu_c, Nt_c = T->Nt_c, am_g = T->am_g, av_g = T->av_g, bv_g = T->bv_g;
    const float *cce = T->cce, *ccg = T->ccg, *cie = T->cie, *cig = T->cig, *cre = T->cre, *crg = T->crg, *cse = T->cse,
                *cge = T->cge, *cgg = T->cgg;
    (void)cce; (void)cse;
    const int kts = 0, kte = nz - 1;

    float tten[KMAX], qvten[KMAX], qcten[KMAX], qiten[KMAX], qrten[KMAX], qsten[KMAX], qgten[KMAX], niten[KMAX], nrten[KMAX];
    float temp[KMAX], qv[KMAX], rc[KMAX], ri[KMAX], rr[KMAX], rs[KMAX], rg[KMAX], ni[KMAX], nr[KMAX];
    float rho[KMAX], rhof[KMAX], lvap[KMAX], ocp[KMAX];
    double ilamg[KMAX], N0_g[KMAX];
    float mvd_r[KMAX], smob[KMAX], smoc[KMAX], xslw_arr[KMAX];
    float sed_r[KMAX], sed_n[KMAX];
    float vtik[KMAX + 1], vtnik[KMAX + 1], vtrk[KMAX + 1], vtnrk[KMAX + 1], vtsk[KMAX + 1], vtgk[KMAX + 1];
    float vts_boost[KMAX];
    int L_qc[KMAX], L_qi[KMAX], L_qr[KMAX], L_qs[KMAX], L_qg[KMAX];
    const float *pres = p1d;

    float rgvm, delta_tp, orho, lfus2, onstep[4];
    double N0_exp, N0_min, lam_exp, lamc, lamr, lamg, lami, ilami;
    float xDc, Dc_b, Dc_g, xDi, xDs, xDg, zeta1, zeta, taud, tau, stoke_g;
    float vti, vtr, vts, vtg, Mrat, ils1, ils2, t1_vts, t2_vts, t3_vts, t4_vts, C_snow;
    float a_, b_, loga_, tf, tempc, tc0, xnc, xri, xni, xmi, oxmi, xrc, xrr, xnr;
    float xsat, rate_max, sump, ratio, clap, fcd, dfcd, otemp, rvs, rvs_p, rvs_pp, gamsc, alphsc, t1_evap, t1_subl;
    float r_frac, g_frac, Ef_rw, Ef_sw, Ef_gw = 0.f, Ef_rr, dtsave, odts, odt, odzq, xslw1, ygra1, zans1;
    int k, n, nstep, idx_tc, idx_t, idx_s, idx_g1, idx_g, idx_r1, idx_r, idx_i1, idx_i, idx_c, idx, ksed1[4];
    int no_micro = 1;
    (void)odt;

    dtsave = dt; odt = 1.f / dt; odts = 1.f / dtsave;

    for (k = 0; k < nz; ++k) { tten[k] = 0; qvten[k] = 0; qcten[k] = 0; qiten[k] = 0; qrten[k] = 0; qsten[k] = 0; qgten[k] = 0; niten[k] = 0; nrten[k] = 0; }
    /* ---- :1240-1319 column -> local arrays ---- */
    for (k = kts; k <= kte; ++k) {
        temp[k] = t1d[k];
        qv[k] = fmaxf(1.E-10f, qv1d[k]);
        rho[k] = 0.622f * pres[k] / (TH_RR2 * temp[k] * (qv[k] + 0.622f));
        if (qc1d[k] > R1) { no_micro = 0; rc[k] = qc1d[k] * rho[k]; L_qc[k] = 1; }
        else { qc1d[k] = 0.0f; rc[k] = R1; L_qc[k] = 0; }
        if (qi1d[k] > R1) {
            no_micro = 0;
            ri[k] = qi1d[k] * rho[k];
            ni[k] = fmaxf(R2, ni1d[k] * rho[k]);
            L_qi[k] = 1;
            lami = d_powf(am_i * cig[1] * T->oig1 * ni[k] / ri[k], T->obmi);
            ilami = 1. / lami;
            xDi = (float)((double)(bm_i + mu_i + 1.f) * ilami);
            if (xDi < 20.E-6f) {
                lami = cie[1] / 20.E-6f;
                ni[k] = (float)fmin(250.e3, (double)(cig[0] * T->oig2 * ri[k] / am_i) * (lami * lami * lami));
            } else if (xDi > 300.E-6f) {
                lami = cie[1] / 300.E-6f;
                ni[k] = (float)((double)(cig[0] * T->oig2 * ri[k] / am_i) * (lami * lami * lami));
            }
        } else { qi1d[k] = 0.0f; ni1d[k] = 0.0f; ri[k] = R1; ni[k] = R2; L_qi[k] = 0; }

        mvd_r[k] = 0.0f;
        if (qr1d[k] > R1) {
            no_micro = 0;
            rr[k] = qr1d[k] * rho[k];
            nr[k] = fmaxf(R2, nr1d[k] * rho[k]);
            L_qr[k] = 1;
            lamr = d_powf(am_r * crg[2] * T->org2 * nr[k] / rr[k], T->obmr);
            mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
            if (mvd_r[k] > 2.5E-3f) {
                mvd_r[k] = 2.5E-3f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * T->org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            } else if (mvd_r[k] < D0r * 0.75f) {
                mvd_r[k] = D0r * 0.75f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * T->org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            }
        } else { qr1d[k] = 0.0f; nr1d[k] = 0.0f; rr[k] = R1; nr[k] = R2; L_qr[k] = 0; }
        if (qs1d[k] > R1) { no_micro = 0; rs[k] = qs1d[k] * rho[k]; L_qs[k] = 1; }
        else { qs1d[k] = 0.0f; rs[k] = R1; L_qs[k] = 0; }
        if (qg1d[k] > R1) { no_micro = 0; rg[k] = qg1d[k] * rho[k]; L_qg[k] = 1; }
        else { qg1d[k] = 0.0f; rg[k] = R1; L_qg[k] = 0; }
    }
    /* ---- no_micro can only be decided after the saturation pass of :1328-1356 ---- */
    if (no_micro) {
        for (k = kts; k <= kte; ++k) {
            const float tc_ = temp[k] - 273.15f;
            const float qvs_ = rslf(pres[k], temp[k]);
            const float qvsi_ = (tc_ <= 0.0f) ? rsif(pres[k], temp[k]) : qvs_;
            float ssati_ = qv[k] / qvsi_ - 1.f;
            if (fabsf(ssati_) < eps) ssati_ = 0.0f;
            if (ssati_ > 0.0f) no_micro = 0;
        }
        if (no_micro) return;     /* :1363 */
    }
    /* ---- :1456-1482 graupel intercept/slope, top-down running minimum ---- */
    N0_min = TH_gonv_max;
    for (k = kte; k >= kts; --k) {
        if (temp[k] < 270.65f && L_qr[k] && mvd_r[k] > 100.E-6f) xslw1 = 4.01f + d_log10f(mvd_r[k]);
        else xslw1 = 0.01f;
        ygra1 = 4.31f + d_log10f(fmaxf(5.E-5f, rg[k]));
        zans1 = 3.1f + (100.f / (300.f * xslw1 * ygra1 / (10.f / xslw1 + 1.f + 0.25f * ygra1) + 30.f + 10.f * ygra1));
        N0_exp = d_pow10f(zans1);
        N0_exp = fmax((double)TH_gonv_min, fmin(N0_exp, (double)TH_gonv_max));
        N0_min = fmin(N0_exp, N0_min);
        N0_exp = N0_min;
        lam_exp = d_pow(N0_exp * am_g * cgg[0] / rg[k], (double)T->oge1);
        lamg = lam_exp * d_powf(cgg[2] * T->ogg2 * T->ogg1, T->obmg);
        ilamg[k] = 1. / lamg;
        N0_g[k] = N0_exp / (cgg[1] * lam_exp) * d_pow(lamg, (double)cge[1]);
    }
    for (k = kts; k <= kte; ++k) {
        double prr_wau = 0, prr_rcw = 0, prr_rcs = 0, prr_rcg = 0, prr_sml = 0, prr_gml = 0, prr_rci = 0, pnr_wau = 0, pnr_rcs = 0, pnr_rcg = 0, pnr_rci = 0, pnr_sml = 0, pnr_gml = 0, pnr_rcr = 0, pnr_rfz = 0, pri_inu = 0, pni_inu = 0, pri_ihm = 0, pni_ihm = 0, pri_wfz = 0, pni_wfz = 0, pri_rfz = 0, pni_rfz = 0, pri_ide = 0, pni_ide = 0, pri_rci = 0, pni_rci = 0, pni_sci = 0, pni_iau = 0, prs_iau = 0, prs_sci = 0, prs_rcs = 0, prs_scw = 0, prs_sde = 0, prs_ihm = 0, prs_ide = 0, prg_scw = 0, prg_rfz = 0, prg_gde = 0, prg_gcw = 0, prg_rci = 0, prg_rcs = 0, prg_rcg = 0, prg_ihm = 0;
        float rhof, rhof2, qvs, qvsi, delQvs, satw, sati, ssatw, ssati, diffu, visco, vsc2, tcond, lvap, ocp, mvd_c;
        float smob = 0.f, smo2 = 0.f, smo1 = 0.f, smo0 = 0.f, smoc = 0.f, smoe = 0.f, smof = 0.f;
        double ilamr, N0_r;
        (void)satw; (void)sati; (void)smo2;
        tempc = temp[k] - 273.15f;
        rhof = sqrtf(TH_rho_not / rho[k]);
        rhof2 = sqrtf(rhof);
        qvs = rslf(pres[k], temp[k]);
        delQvs = fmaxf(0.0f, rslf(pres[k], 273.15f) - qv[k]);
        if (tempc <= 0.0f) qvsi = rsif(pres[k], temp[k]); else qvsi = qvs;
        satw = qv[k] / qvs;
        sati = qv[k] / qvsi;
        ssatw = satw - 1.f;
        ssati = sati - 1.f;
        if (fabsf(ssatw) < eps) ssatw = 0.0f;
        if (fabsf(ssati) < eps) ssati = 0.0f;
        diffu = 2.11E-5f * d_powf(temp[k] / 273.15f, 1.94f) * (101325.f / pres[k]);
        if (tempc >= 0.0f) visco = (1.718f + 0.0049f * tempc) * 1.0E-5f;
        else visco = (1.718f + 0.0049f * tempc - 1.2E-5f * tempc * tempc) * 1.0E-5f;
        ocp = 1.f / (TH_Cp2 * (1.f + 0.887f * qv[k]));
        vsc2 = sqrtf(rho[k] / visco);
        lvap = lvap0 + (2106.0f - 4218.0f) * tempc;
        tcond = (5.69f + 0.0168f * tempc) * 1.0E-5f * 418.936f;
            if (L_qs[k]) {
        tc0 = fminf(-0.1f, temp[k] - 273.15f);
        smob = rs[k] * T->oams;
        if (TH_bm_s > (2.0f - 1.e-3f) && TH_bm_s < (2.0f + 1.e-3f)) smo2 = smob;
        else {
            loga_ = snow_poly_f(sa, tc0, TH_bm_s); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, TH_bm_s);
            smo2 = d_powf(smob / a_, 1.f / b_);
        }
        loga_ = sa[0] + sa[1] * tc0 + sa[4] * tc0 * tc0 + sa[8] * tc0 * tc0 * tc0;
        a_ = d_pow10f(loga_);
        b_ = sb[0] + sb[1] * tc0 + sb[4] * tc0 * tc0 + sb[8] * tc0 * tc0 * tc0;
        smo0 = a_ * d_powf(smo2, b_);
        loga_ = sa[0] + sa[1] * tc0 + sa[2] + sa[3] * tc0 + sa[4] * tc0 * tc0 + sa[5] + sa[6] * tc0 * tc0 + sa[7] * tc0
              + sa[8] * tc0 * tc0 * tc0 + sa[9];
        a_ = d_pow10f(loga_);
        b_ = sb[0] + sb[1] * tc0 + sb[2] + sb[3] * tc0 + sb[4] * tc0 * tc0 + sb[5] + sb[6] * tc0 * tc0 + sb[7] * tc0
           + sb[8] * tc0 * tc0 * tc0 + sb[9];
        smo1 = a_ * d_powf(smo2, b_);
        loga_ = snow_poly_f(sa, tc0, T->cse[0]); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, T->cse[0]);
        smoc = a_ * d_powf(smo2, b_);
        loga_ = snow_poly_f(sa, tc0, T->cse[12]); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, T->cse[12]);
        smoe = a_ * d_powf(smo2, b_);
        loga_ = snow_poly_f(sa, tc0, T->cse[15]); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, T->cse[15]);
        smof = a_ * d_powf(smo2, b_);
            }
        lamr = d_powf(am_r * crg[2] * T->org2 * nr[k] / rr[k], T->obmr);
        ilamr = 1. / lamr;
        mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
        N0_r = (double)(nr[k] * T->org2) * d_pow(lamr, (double)cre[1]);
            if (L_qr[k] && mvd_r[k] > D0r) {
            Ef_rr = 2.0f - d_expf(2300.0f * (mvd_r[k] - 1600.0E-6f));
            pnr_rcr = Ef_rr * 4.f * nr[k] * rr[k];
        }
        mvd_c = D0c;
        if (L_qc[k]) {
        xDc = fmaxf(D0c * 1.E6f, (d_powf(rc[k] / (am_r * Nt_c), T->obmr)) * 1.E6f);
        lamc = d_powf(Nt_c * am_r * ccg[1] * T->ocg1 / rc[k], T->obmr);
        mvd_c = (float)((double)(3.0f + mu_c + 0.672f) / lamc);
        if (rc[k] > 0.01e-3f) {
            Dc_g = (float)(((double)d_powf(ccg[2] * T->ocg2, T->obmr) / lamc) * (double)1.E6f);
            Dc_b = d_powf(xDc * xDc * xDc * Dc_g * Dc_g * Dc_g - xDc * xDc * xDc * xDc * xDc * xDc, 1.f / 6.f);
            zeta1 = 0.5f * ((6.25E-6f * xDc * Dc_b * Dc_b * Dc_b - 0.4f) + fabsf(6.25E-6f * xDc * Dc_b * Dc_b * Dc_b - 0.4f));
            zeta = 0.027f * rc[k] * zeta1;
            taud = 0.5f * ((0.5f * Dc_b - 7.5f) + fabsf(0.5f * Dc_b - 7.5f)) + R1;
            tau = 3.72f / (rc[k] * taud);
            prr_wau = zeta / tau;
            prr_wau = fmin((double)(rc[k] * odts), prr_wau);
            pnr_wau = prr_wau / (double)(am_r * mu_c * D0r * D0r * D0r);
        }
        if (L_qr[k] && mvd_r[k] > D0r && mvd_c > D0c) {
            lamr = 1. / ilamr;
            idx = 1 + (int)(NBINS * log((double)mvd_r[k] / T->Dr[0]) / log(T->Dr[NBINS - 1] / T->Dr[0]));
            idx = imin(idx, NBINS);
            int ic = (int)(mvd_c * 1.E6f);
            ic = imax(1, imin(ic, NBINS));          /* the reference does not bound this index */
            Ef_rw = (float)T->t_Efrw[(idx - 1) + NBINS * (ic - 1)];
            prr_rcw = (double)(rhof * T->t1_qr_qc * Ef_rw * rc[k]) * N0_r * d_pow(lamr + (double)fv_r, -(double)cre[8]);
            prr_rcw = fmin((double)(rc[k] * odts), prr_rcw);
        }
            }
        vts_boost[k] = 1.5f;
        tempc = temp[k] - 273.15f;
        idx_tc = imax(1, imin((int)lroundf(-tempc), 45));
        idx_t = (int)((tempc - 2.5f) / 5.f) - 1;
        idx_t = imax(1, -idx_t);
        idx_t = imin(idx_t, NTB_T);

        if (rc[k] > T->r_c[0]) { idx_c = dec_index_f(rc[k], T->nic2); idx_c = imax(1, imin(idx_c, NTB_C)); } else idx_c = 1;
        if (ri[k] > T->r_i[0]) { idx_i = dec_index_f(ri[k], T->nii2); idx_i = imax(1, imin(idx_i, NTB_I)); } else idx_i = 1;
        if (ni[k] > T->Nt_i[0]) { idx_i1 = dec_index_f(ni[k], T->nii3); idx_i1 = imax(1, imin(idx_i1, NTB_I1)); } else idx_i1 = 1;
        if (rr[k] > T->r_r[0]) {
            idx_r = dec_index_f(rr[k], T->nir2); idx_r = imax(1, imin(idx_r, NTB_R));
            lamr = 1. / ilamr;
            lam_exp = lamr * cube_f(crg[2] * T->org2 * T->org1);
            N0_exp = (double)(T->org1 * rr[k] / am_r) * d_pow(lam_exp, (double)cre[0]);
            idx_r1 = dec_index_d(N0_exp, T->nir3); idx_r1 = imax(1, imin(idx_r1, NTB_R1));
        } else { idx_r = 1; idx_r1 = NTB_R1; }
        if (rs[k] > T->r_s[0]) { idx_s = dec_index_f(rs[k], T->nis2); idx_s = imax(1, imin(idx_s, NTB_S)); } else idx_s = 1;
        if (rg[k] > T->r_g[0]) {
            idx_g = dec_index_f(rg[k], T->nig2); idx_g = imax(1, imin(idx_g, NTB_G));
            lamg = 1. / ilamg[k];
            lam_exp = lamg * cube_f(cgg[2] * T->ogg2 * T->ogg1);
            N0_exp = (double)(T->ogg1 * rg[k] / am_g) * d_pow(lam_exp, (double)cge[0]);
            idx_g1 = dec_index_d(N0_exp, T->nig3); idx_g1 = imax(1, imin(idx_g1, NTB_G1));
        } else { idx_g = 1; idx_g1 = NTB_G1; }

        /* deposition/sublimation prefactor :1679-1695 */
        otemp = 1.f / temp[k];
        rvs = rho[k] * qvsi;
        rvs_p = rvs * otemp * (lsub * otemp * oRv - 1.f);
        rvs_pp = rvs * (otemp * (lsub * otemp * oRv - 1.f) * otemp * (lsub * otemp * oRv - 1.f)
                        + (-2.f * lsub * otemp * otemp * otemp * oRv) + otemp * otemp);
        gamsc = lsub * diffu / tcond * rvs_p;
        alphsc = 0.5f * (gamsc / (1.f + gamsc)) * (gamsc / (1.f + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
        alphsc = fmaxf(1.E-9f, alphsc);
        xsat = ssati;
        if (fabsf(xsat) < 1.E-9f) xsat = 0.f;
        t1_subl = 4.f * PI2 * (1.0f - alphsc * xsat + 2.f * alphsc * alphsc * xsat * xsat
                               - 5.f * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1.f + gamsc);

        /* snow / graupel collecting cloud water :1698-1725 */
        if (L_qc[k] && mvd_c > D0c) {
            xDs = 0.0f;
            if (L_qs[k]) xDs = smoc / smob;
            if (xDs > D0s) {
                idx = 1 + (int)(NBINS * log((double)xDs / T->Ds[0]) / log(T->Ds[NBINS - 1] / T->Ds[0]));
                idx = imin(idx, NBINS);
                int ic = (int)(mvd_c * 1.E6f); ic = imax(1, imin(ic, NBINS));
                Ef_sw = (float)T->t_Efsw[(idx - 1) + NBINS * (ic - 1)];
                prs_scw = rhof * T->t1_qs_qc * Ef_sw * rc[k] * smoe;
            }
            if (rg[k] >= T->r_g[0] && mvd_c > D0c) {
                xDg = (float)((double)(bm_g + mu_g + 1.f) * ilamg[k]);
                vtg = (float)((double)(rhof * av_g * cgg[5] * T->ogg3) * d_pow(ilamg[k], (double)bv_g));
                stoke_g = mvd_c * mvd_c * vtg * rho_w / (9.f * visco * xDg);
                if (xDg > D0g) {
                    if (stoke_g >= 0.4f && stoke_g <= 10.f) Ef_gw = 0.55f * d_log10f(2.51f * stoke_g);
                    else if (stoke_g < 0.4f) Ef_gw = 0.0f;
                    else if (stoke_g > 10.f) Ef_gw = 0.77f;
                    prg_gcw = (double)(rhof * T->t1_qg_qc * Ef_gw * rc[k]) * N0_g[k] * d_pow(ilamg[k], (double)cge[8]);
                }
            }
        }

        /* rain collecting snow / graupel :1730-1783 */
        if (rr[k] >= T->r_r[0]) {
            if (rs[k] >= T->r_s[0]) {
                if (temp[k] < T_0) {
                    prr_rcs = -(T4S(tmr_racs2) + T4S(tcr_sacr2) + T4S(tmr_racs1) + T4S(tcr_sacr1));
                    prs_rcs = T4S(tmr_racs2) + T4S(tcr_sacr2) - T4S(tcs_racs1) - T4S(tms_sacr1);
                    prg_rcs = T4S(tmr_racs1) + T4S(tcr_sacr1) + T4S(tcs_racs1) + T4S(tms_sacr1);
                    prr_rcs = fmax((double)(-rr[k] * odts), prr_rcs);
                    prs_rcs = fmax((double)(-rs[k] * odts), prs_rcs);
                    prg_rcs = fmin((double)((rr[k] + rs[k]) * odts), prg_rcs);
                    pnr_rcs = T4S(tnr_racs1) + T4S(tnr_racs2) + T4S(tnr_sacr1) + T4S(tnr_sacr2);
                } else {
                    prs_rcs = -T4S(tcs_racs1) - T4S(tms_sacr1) + T4S(tmr_racs2) + T4S(tcr_sacr2);
                    prs_rcs = fmax((double)(-rs[k] * odts), prs_rcs);
                    prr_rcs = -prs_rcs;
                    pnr_rcs = T4S(tnr_racs2) + T4S(tnr_sacr2);
                }
                pnr_rcs = fmin((double)(nr[k] * odts), pnr_rcs);
            }
            if (rg[k] >= T->r_g[0]) {
                if (temp[k] < T_0) {
                    prg_rcg = T4G(tmr_racg) + T4G(tcr_gacr);
                    prg_rcg = fmin((double)(rr[k] * odts), prg_rcg);
                    prr_rcg = -prg_rcg;
                    pnr_rcg = T4G(tnr_racg) + T4G(tnr_gacr);
                    pnr_rcg = fmin((double)(nr[k] * odts), pnr_rcg);
                } else {
                    prr_rcg = T4G(tcg_racg);
                    prr_rcg = fmin((double)(rg[k] * odts), prr_rcg);
                    prg_rcg = -prr_rcg;
                }
            }
        }

        if (temp[k] < T_0) {      /* :1789-1949 sub-zero processes */
            vts_boost[k] = 1.0f;
            rate_max = (qv[k] - qvsi) * rho[k] * odts * 0.999f;
            if (rr[k] > T->r_r[0]) {
                prg_rfz = T3R(tpg_qrfz) * odts;
                pri_rfz = T3R(tpi_qrfz) * odts;
                pni_rfz = T3R(tni_qrfz) * odts;
                pnr_rfz = T3R(tnr_qrfz) * odts;
                pnr_rfz = fmin((double)(nr[k] * odts), pnr_rfz);
            } else if (rr[k] > R1 && temp[k] < HGFR) {
                pri_rfz = rr[k] * odts;
                pnr_rfz = nr[k] * odts;
                pni_rfz = pnr_rfz;
            }
            if (rc[k] > T->r_c[0]) {
                pri_wfz = T2C(tpi_qcfz) * odts;
                pri_wfz = fmin((double)(rc[k] * odts), pri_wfz);
                pni_wfz = T2C(tni_qcfz) * odts;
                pni_wfz = fmin(fmin((double)(Nt_c * odts), pri_wfz / (double)(2.f * xm0i)), pni_wfz);
            } else if (rc[k] > R1 && temp[k] < HGFR) {
                pri_wfz = rc[k] * odts;
                pni_wfz = fmin(fmin((double)(Nt_c * odts), pri_wfz / (double)(2.f * xm0i)), pni_wfz);
            }
            if ((ssati >= 0.25f) || (ssatw > eps && temp[k] < 261.15f)) {
                xnc = fminf(250.E3f, T->TNO * d_expf(TH_ATO * (T_0 - temp[k])));
                xni = (float)((double)ni[k] + (pni_rfz + pni_wfz) * (double)dtsave);
                pni_inu = 0.5f * (xnc - xni + fabsf(xnc - xni)) * odts;
                pri_inu = fmin((double)rate_max, (double)xm0i * pni_inu);
                pni_inu = pri_inu / (double)xm0i;
            }
            if (L_qi[k]) {
                lami = d_powf(am_i * cig[1] * T->oig1 * ni[k] / ri[k], T->obmi);
                ilami = 1. / lami;
                xDi = (float)fmax((double)T->D0i, (double)(bm_i + mu_i + 1.f) * ilami);
                xmi = am_i * (xDi * xDi * xDi);
                oxmi = 1.f / xmi;
                pri_ide = (double)(C_cube * t1_subl * diffu * ssati * rvs * T->oig1 * cig[4] * ni[k]) * ilami;
                if (pri_ide < 0.0) {
                    pri_ide = fmax(fmax((double)(-ri[k] * odts), pri_ide), (double)rate_max);
                    pni_ide = pri_ide * (double)oxmi;
                    pni_ide = fmax((double)(-ni[k] * odts), pni_ide);
                } else {
                    pri_ide = fmin(pri_ide, (double)rate_max);
                    prs_ide = (1.0 - T2I(tpi_ide)) * pri_ide;
                    pri_ide = T2I(tpi_ide) * pri_ide;
                }
                if ((idx_i == NTB_I) || (xDi > 5.0f * D0s)) {
                    prs_iau = ri[k] * .99f * odts;
                    pni_iau = ni[k] * .95f * odts;
                } else if (xDi < 0.1f * D0s) {
                    prs_iau = 0.; pni_iau = 0.;
                } else {
                    prs_iau = T2I(tps_iaus) * odts;
                    prs_iau = fmin((double)(ri[k] * .99f * odts), prs_iau);
                    pni_iau = T2I(tni_iaus) * odts;
                    pni_iau = fmin((double)(ni[k] * .95f * odts), pni_iau);
                }
            }
            if (L_qs[k]) {
                C_snow = T->C_sqrd + (tempc + 15.f) * (T->C_cubes - T->C_sqrd) / (-30.f + 15.f);
                C_snow = fmaxf(T->C_sqrd, fminf(C_snow, T->C_cubes));
                prs_sde = C_snow * t1_subl * diffu * ssati * rvs
                             * (T->t1_qs_sd * smo1 + T->t2_qs_sd * rhof2 * vsc2 * smof);
                if (prs_sde < 0.) prs_sde = fmax(fmax((double)(-rs[k] * odts), prs_sde), (double)rate_max);
                else prs_sde = fmin(prs_sde, (double)rate_max);
            }
            if (L_qg[k] && ssati < -eps) {
                prg_gde = (double)(C_cube * t1_subl * diffu * ssati * rvs) * N0_g[k]
                             * ((double)T->t1_qg_sd * d_pow(ilamg[k], (double)cge[9])
                                + (double)(T->t2_qg_sd * vsc2 * rhof2) * d_pow(ilamg[k], (double)cge[10]));
                if (prg_gde < 0.) prg_gde = fmax(fmax((double)(-rg[k] * odts), prg_gde), (double)rate_max);
                else prg_gde = fmin(prg_gde, (double)rate_max);
            }
            if (L_qi[k]) {
                lami = d_powf(am_i * cig[1] * T->oig1 * ni[k] / ri[k], T->obmi);
                ilami = 1. / lami;
                xDi = (float)fmax((double)T->D0i, (double)(bm_i + mu_i + 1.f) * ilami);
                xmi = am_i * (xDi * xDi * xDi);
                oxmi = 1.f / xmi;
                if (rs[k] >= T->r_s[0]) {
                    prs_sci = T->t1_qs_qi * rhof * T->Ef_si * ri[k] * smoe;
                    pni_sci = prs_sci * (double)oxmi;
                }
                if (rr[k] >= T->r_r[0] && mvd_r[k] > 4.f * xDi) {
                    lamr = 1. / ilamr;
                    pri_rci = (double)(rhof * T->t1_qr_qi * T->Ef_ri * ri[k]) * N0_r * d_pow(lamr + (double)fv_r, -(double)cre[8]);
                    pnr_rci = (double)(rhof * T->t1_qr_qi * T->Ef_ri * ni[k]) * N0_r * d_pow(lamr + (double)fv_r, -(double)cre[8]);
                    pni_rci = pri_rci * (double)oxmi;
                    prr_rci = (double)(rhof * T->t2_qr_qi * T->Ef_ri * ni[k]) * N0_r * d_pow(lamr + (double)fv_r, -(double)cre[7]);
                    prr_rci = fmin((double)(rr[k] * odts), prr_rci);
                    prg_rci = pri_rci + prr_rci;
                }
            }
            if (prg_gcw > (double)eps && tempc > -8.0f) {
                tf = 0.f;
                if (tempc >= -5.0f && tempc < -3.0f) tf = 0.5f * (-3.0f - tempc);
                else if (tempc > -8.0f && tempc < -5.0f) tf = 0.33333333f * (8.0f + tempc);
                pni_ihm = (double)(3.5E8f * tf) * prg_gcw;
                pri_ihm = (double)xm0i * pni_ihm;
                prs_ihm = prs_scw / (prs_scw + prg_gcw) * pri_ihm;
                prg_ihm = prg_gcw / (prs_scw + prg_gcw) * pri_ihm;
            }
            if (prs_scw > (double)5.0f * prs_sde && prs_sde > (double)eps) {
                r_frac = (float)fmin(30.0, prs_scw / prs_sde);
                g_frac = fminf(0.75f, 0.05f + (r_frac - 5.f) * .028f);
                vts_boost[k] = fminf(1.5f, 1.1f + (r_frac - 5.f) * .016f);
                prg_scw = (double)g_frac * prs_scw;
                prs_scw = (double)(1.f - g_frac) * prs_scw;
            }
        } else {                  /* :1953-2005 melting */
            if (L_qs[k]) {
                prr_sml = (tempc * tcond - lvap0 * diffu * delQvs)
                             * (T->t1_qs_me * smo1 + T->t2_qs_me * rhof2 * vsc2 * smof);
                prr_sml = prr_sml + (double)(4218.f * olfus * tempc) * (prr_rcs + prs_scw);
                prr_sml = fmin((double)(rs[k] * odts), fmax(0., prr_sml));
                pnr_sml = (double)(smo0 / rs[k]) * prr_sml * (double)d_pow10f(-0.75f * tempc);
                pnr_sml = fmin((double)(smo0 * odts), pnr_sml);
                if (tempc > 3.5f || rs[k] < 0.005E-3f) pnr_sml = 0.0;
                if (ssati < 0.f) {
                    prs_sde = T->C_cubes * t1_subl * diffu * ssati * rvs
                                 * (T->t1_qs_sd * smo1 + T->t2_qs_sd * rhof2 * vsc2 * smof);
                    prs_sde = fmax((double)(-rs[k] * odts), prs_sde);
                }
            }
            if (L_qg[k]) {
                prr_gml = (double)(tempc * tcond - lvap0 * diffu * delQvs) * N0_g[k]
                             * ((double)T->t1_qg_me * d_pow(ilamg[k], (double)cge[9])
                                + (double)(T->t2_qg_me * rhof2 * vsc2) * d_pow(ilamg[k], (double)cge[10]));
                prr_gml = fmin((double)(rg[k] * odts), fmax(0., prr_gml));
                pnr_gml = N0_g[k] * (double)cgg[1] * d_pow(ilamg[k], (double)cge[1]) / (double)rg[k]
                             * prr_gml * (double)d_pow10f(-1.5f * tempc);
                if (tempc > 7.5f || rg[k] < 0.005E-3f) pnr_gml = 0.0;
                if (ssati < 0.f) {
                    prg_gde = (double)(C_cube * t1_subl * diffu * ssati * rvs) * N0_g[k]
                                 * ((double)T->t1_qg_sd * d_pow(ilamg[k], (double)cge[9])
                                    + (double)(T->t2_qg_sd * vsc2 * rhof2) * d_pow(ilamg[k], (double)cge[10]));
                    prg_gde = fmax((double)(-rg[k] * odts), prg_gde);
                }
            }
            if (dt > 120.f) {
                prr_rcw = prr_rcw + prs_scw + prg_gcw;
                prs_scw = 0.; prg_gcw = 0.;
            }
        }
            sump = (float)(pri_inu + pri_ide + prs_ide + prs_sde + prg_gde);
        rate_max = (qv[k] - qvsi) * odts * 0.999f;
        if ((sump > eps && sump > rate_max) || (sump < -eps && sump < rate_max)) {
            ratio = rate_max / sump;
            pri_inu *= ratio; pri_ide *= ratio; pni_ide *= ratio; prs_ide *= ratio; prs_sde *= ratio; prg_gde *= ratio;
        }
        sump = (float)(-prr_wau - pri_wfz - prr_rcw - prs_scw - prg_scw - prg_gcw);
        rate_max = -rc[k] * odts;
        if (sump < rate_max && L_qc[k]) {
            ratio = rate_max / sump;
            prr_wau *= ratio; pri_wfz *= ratio; prr_rcw *= ratio; prs_scw *= ratio; prg_scw *= ratio; prg_gcw *= ratio;
        }
        sump = (float)(pri_ide - prs_iau - prs_sci - pri_rci);
        rate_max = -ri[k] * odts;
        if (sump < rate_max && L_qi[k]) {
            ratio = rate_max / sump;
            pri_ide *= ratio; prs_iau *= ratio; prs_sci *= ratio; pri_rci *= ratio;
        }
        sump = (float)(-prg_rfz - pri_rfz - prr_rci + prr_rcs + prr_rcg);
        rate_max = -rr[k] * odts;
        if (sump < rate_max && L_qr[k]) {
            ratio = rate_max / sump;
            prg_rfz *= ratio; pri_rfz *= ratio; prr_rci *= ratio; prr_rcs *= ratio; prr_rcg *= ratio;
        }
        sump = (float)(prs_sde - prs_ihm - prr_sml + prs_rcs);
        rate_max = -rs[k] * odts;
        if (sump < rate_max && L_qs[k]) {
            ratio = rate_max / sump;
            prs_sde *= ratio; prs_ihm *= ratio; prr_sml *= ratio; prs_rcs *= ratio;
        }
        sump = (float)(prg_gde - prg_ihm - prr_gml + prg_rcg);
        rate_max = -rg[k] * odts;
        if (sump < rate_max && L_qg[k]) {
            ratio = rate_max / sump;
            prg_gde *= ratio; prg_ihm *= ratio; prr_gml *= ratio; prg_rcg *= ratio;
        }
        pri_ihm = prs_ihm + prg_ihm;
        ratio = (float)fmin(fabs(prr_rcg), fabs(prg_rcg));
        prr_rcg = ratio * copysignf(1.0f, (float)prr_rcg);
        prg_rcg = -prr_rcg;
        if (temp[k] > T_0) {
            ratio = (float)fmin(fabs(prr_rcs), fabs(prs_rcs));
            prr_rcs = ratio * copysignf(1.0f, (float)prr_rcs);
            prs_rcs = -prr_rcs;
        }
            orho = 1.f / rho[k];
        lfus2 = lsub - lvap;
        qvten[k] = (float)(qvten[k] + (-pri_inu - pri_ide - prs_ide - prs_sde - prg_gde) * orho);
        qcten[k] = (float)(qcten[k] + (-prr_wau - pri_wfz - prr_rcw - prs_scw - prg_scw - prg_gcw) * orho);
        qiten[k] = (float)(qiten[k] + (pri_inu + pri_ihm + pri_wfz + pri_rfz + pri_ide
                                       - prs_iau - prs_sci - pri_rci) * orho);
        niten[k] = (float)(niten[k] + (pni_inu + pni_ihm + pni_wfz + pni_rfz + pni_ide
                                       - pni_iau - pni_sci - pni_rci) * orho);
        xri = fmaxf(R1, (qi1d[k] + qiten[k] * dtsave) * rho[k]);
        xni = fmaxf(R2, (ni1d[k] + niten[k] * dtsave) * rho[k]);
        if (xri > R1) {
            lami = d_powf(am_i * cig[1] * T->oig1 * xni / xri, T->obmi);
            ilami = 1. / lami;
            xDi = (float)((double)(bm_i + mu_i + 1.f) * ilami);
            if (xDi < 20.E-6f) {
                lami = cie[1] / 20.E-6f;
                xni = (float)fmin(250.e3, (double)(cig[0] * T->oig2 * xri / am_i) * (lami * lami * lami));
                niten[k] = (xni - ni1d[k] * rho[k]) * odts * orho;
            } else if (xDi > 300.E-6f) {
                lami = cie[1] / 300.E-6f;
                xni = (float)((double)(cig[0] * T->oig2 * xri / am_i) * (lami * lami * lami));
                niten[k] = (xni - ni1d[k] * rho[k]) * odts * orho;
            }
        } else niten[k] = -ni1d[k] * odts;
        xni = fmaxf(0.f, (ni1d[k] + niten[k] * dtsave) * rho[k]);
        if (xni > 250.E3f) niten[k] = (250.E3f - ni1d[k] * rho[k]) * odts * orho;

        qrten[k] = (float)(qrten[k] + (prr_wau + prr_rcw + prr_sml + prr_gml + prr_rcs + prr_rcg
                                       - prg_rfz - pri_rfz - prr_rci) * orho);
        nrten[k] = (float)(nrten[k] + (pnr_wau + pnr_sml + pnr_gml
                                       - (pnr_rfz + pnr_rcr + pnr_rcg + pnr_rcs + pnr_rci)) * orho);
        xrr = fmaxf(R1, (qr1d[k] + qrten[k] * dtsave) * rho[k]);
        xnr = fmaxf(R2, (nr1d[k] + nrten[k] * dtsave) * rho[k]);
        if (xrr > R1) {
            lamr = d_powf(am_r * crg[2] * T->org2 * xnr / xrr, T->obmr);
            mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
            if (mvd_r[k] > 2.5E-3f) {
                mvd_r[k] = 2.5E-3f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                xnr = (float)((double)(crg[1] * T->org3 * xrr) * (lamr * lamr * lamr) / (double)am_r);
                nrten[k] = (xnr - nr1d[k] * rho[k]) * odts * orho;
            } else if (mvd_r[k] < D0r * 0.75f) {
                mvd_r[k] = D0r * 0.75f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                xnr = (float)((double)(crg[1] * T->org3 * xrr) * (lamr * lamr * lamr) / (double)am_r);
                nrten[k] = (xnr - nr1d[k] * rho[k]) * odts * orho;
            }
        } else { qrten[k] = -qr1d[k] * odts; nrten[k] = -nr1d[k] * odts; }

        qsten[k] = (float)(qsten[k] + (prs_iau + prs_sde + prs_sci + prs_scw + prs_rcs + prs_ide
                                       - prs_ihm - prr_sml) * orho);
        qgten[k] = (float)(qgten[k] + (prg_scw + prg_rfz + prg_gde + prg_rcg + prg_gcw + prg_rci
                                       + prg_rcs - prg_ihm - prr_gml) * orho);
        if (temp[k] < T_0) {
            tten[k] = (float)(tten[k] + ((double)(lsub * ocp) * (pri_inu + pri_ide + prs_ide + prs_sde + prg_gde)
                              + (double)(lfus2 * ocp) * (pri_wfz + pri_rfz + prg_rfz + prs_scw + prg_scw + prg_gcw
                                                             + prg_rcs + prs_rcs + prr_rci + prg_rcg)) * orho * 1);
        } else {
            tten[k] = (float)(tten[k] + ((double)(TH_lfus * ocp) * (-prr_sml - prr_gml - prr_rcg - prr_rcs)
                              + (double)(lsub * ocp) * (prs_sde + prg_gde)) * orho * 1);
        }
        }
    for (k = kts; k <= kte; ++k) {
        float rhof, rhof2, qvs, ssatw, diffu, visco, vsc2, tcond, lvt2, smo2 = 0.f, smod = 0.f;
        double ilamr, N0_r, prw_vcd = 0, prv_rev = 0, pnr_rev = 0;
        (void)rhof2; (void)smod; (void)prv_rev;
        temp[k] = t1d[k] + dt * tten[k];
        otemp = 1.f / temp[k];
        tempc = temp[k] - 273.15f;
        qv[k] = fmaxf(1.E-10f, qv1d[k] + dt * qvten[k]);
        rho[k] = 0.622f * pres[k] / (TH_RR2 * temp[k] * (qv[k] + 0.622f));
        rhof = sqrtf(TH_rho_not / rho[k]);
        rhof2 = sqrtf(rhof);
        qvs = rslf(pres[k], temp[k]);
        ssatw = qv[k] / qvs - 1.f;
        if (fabsf(ssatw) < eps) ssatw = 0.0f;
        diffu = 2.11E-5f * d_powf(temp[k] / 273.15f, 1.94f) * (101325.f / pres[k]);
        if (tempc >= 0.0f) visco = (1.718f + 0.0049f * tempc) * 1.0E-5f;
        else visco = (1.718f + 0.0049f * tempc - 1.2E-5f * tempc * tempc) * 1.0E-5f;
        vsc2 = sqrtf(rho[k] / visco);
        lvap[k] = lvap0 + (2106.0f - 4218.0f) * tempc;
        tcond = (5.69f + 0.0168f * tempc) * 1.0E-5f * 418.936f;
        ocp[k] = 1.f / (TH_Cp2 * (1.f + 0.887f * qv[k]));
        lvt2 = lvap[k] * lvap[k] * ocp[k] * oRv * otemp * otemp;

        if ((qc1d[k] + qcten[k] * dt) > R1) { rc[k] = (qc1d[k] + qcten[k] * dt) * rho[k]; L_qc[k] = 1; }
        else { rc[k] = R1; L_qc[k] = 0; }
        if ((qi1d[k] + qiten[k] * dt) > R1) {
            ri[k] = (qi1d[k] + qiten[k] * dt) * rho[k];
            ni[k] = fmaxf(R2, (ni1d[k] + niten[k] * dt) * rho[k]);
            L_qi[k] = 1;
        } else { ri[k] = R1; ni[k] = R2; L_qi[k] = 0; }
        if ((qr1d[k] + qrten[k] * dt) > R1) {
            rr[k] = (qr1d[k] + qrten[k] * dt) * rho[k];
            nr[k] = fmaxf(R2, (nr1d[k] + nrten[k] * dt) * rho[k]);
            L_qr[k] = 1;
            lamr = d_powf(am_r * crg[2] * T->org2 * nr[k] / rr[k], T->obmr);
            mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
            if (mvd_r[k] > 2.5E-3f) {
                mvd_r[k] = 2.5E-3f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * T->org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            } else if (mvd_r[k] < D0r * 0.75f) {
                mvd_r[k] = D0r * 0.75f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * T->org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            }
        } else { rr[k] = R1; nr[k] = R2; L_qr[k] = 0; }
        if ((qs1d[k] + qsten[k] * dt) > R1) { rs[k] = (qs1d[k] + qsten[k] * dt) * rho[k]; L_qs[k] = 1; }
        else { rs[k] = R1; L_qs[k] = 0; }
        if ((qg1d[k] + qgten[k] * dt) > R1) { rg[k] = (qg1d[k] + qgten[k] * dt) * rho[k]; L_qg[k] = 1; }
        else { rg[k] = R1; L_qg[k] = 0; }
            if (L_qs[k]) {
        tc0 = fminf(-0.1f, temp[k] - 273.15f);
        smob[k] = rs[k] * T->oams;
        if (TH_bm_s > (2.0f - 1.e-3f) && TH_bm_s < (2.0f + 1.e-3f)) smo2 = smob[k];
        else {
            loga_ = snow_poly_f(sa, tc0, TH_bm_s); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, TH_bm_s);
            smo2 = d_powf(smob[k] / a_, 1.f / b_);
        }
        loga_ = snow_poly_f(sa, tc0, T->cse[0]); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, T->cse[0]);
        smoc[k] = a_ * d_powf(smo2, b_);
        loga_ = snow_poly_f(sa, tc0, T->cse[13]); a_ = d_pow10f(loga_); b_ = snow_poly_f(sb, tc0, T->cse[13]);
        smod = a_ * d_powf(smo2, b_);
            }
        /* input of the second graupel chain (:2381-2385) must see the TAU+1 state of THIS point in the sequence */
        xslw_arr[k] = (temp[k] < 270.65f && L_qr[k] && mvd_r[k] > 100.E-6f) ? 4.01f + d_log10f(mvd_r[k]) : 0.01f;
        lamr = d_powf(am_r * crg[2] * T->org2 * nr[k] / rr[k], T->obmr);
        ilamr = 1. / lamr;
        mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
        N0_r = (double)(nr[k] * T->org2) * d_pow(lamr, (double)cre[1]);
            if ((ssatw > eps) || (ssatw < -eps && L_qc[k])) {
            clap = (qv[k] - qvs) / (1.f + lvt2 * qvs);
            for (n = 1; n <= 3; ++n) {
                fcd = qvs * d_expf(lvt2 * clap) - qv[k] + clap;
                dfcd = qvs * lvt2 * d_expf(lvt2 * clap) + 1.f;
                clap = clap - fcd / dfcd;
            }
            xrc = rc[k] + clap;
            if (xrc > 0.0f) prw_vcd = clap * odt;
            else prw_vcd = -rc[k] / rho[k] * odts;
            qcten[k] = (float)(qcten[k] + prw_vcd);
            qvten[k] = (float)(qvten[k] - prw_vcd);
            tten[k] = (float)(tten[k] + (double)(lvap[k] * ocp[k]) * prw_vcd * 1);
            rc[k] = fmaxf(R1, (qc1d[k] + dt * qcten[k]) * rho[k]);
            qv[k] = fmaxf(1.E-10f, qv1d[k] + dt * qvten[k]);
            temp[k] = t1d[k] + dt * tten[k];
            rho[k] = 0.622f * pres[k] / (TH_RR2 * temp[k] * (qv[k] + 0.622f));
            qvs = rslf(pres[k], temp[k]);
            ssatw = qv[k] / qvs - 1.f;
        }
            if ((ssatw < -eps) && L_qr[k] && (!(prw_vcd > 0.))) {
            tempc = temp[k] - 273.15f;
            otemp = 1.f / temp[k];
            rhof = sqrtf(TH_rho_not / rho[k]);
            rhof2 = sqrtf(rhof);
            diffu = 2.11E-5f * d_powf(temp[k] / 273.15f, 1.94f) * (101325.f / pres[k]);
            if (tempc >= 0.0f) visco = (1.718f + 0.0049f * tempc) * 1.0E-5f;
            else visco = (1.718f + 0.0049f * tempc - 1.2E-5f * tempc * tempc) * 1.0E-5f;
            vsc2 = sqrtf(rho[k] / visco);
            lvap[k] = lvap0 + (2106.0f - 4218.0f) * tempc;
            tcond = (5.69f + 0.0168f * tempc) * 1.0E-5f * 418.936f;
            ocp[k] = 1.f / (TH_Cp2 * (1.f + 0.887f * qv[k]));
            rvs = rho[k] * qvs;
            rvs_p = rvs * otemp * (lvap[k] * otemp * oRv - 1.f);
            rvs_pp = rvs * (otemp * (lvap[k] * otemp * oRv - 1.f) * otemp * (lvap[k] * otemp * oRv - 1.f)
                            + (-2.f * lvap[k] * otemp * otemp * otemp * oRv) + otemp * otemp);
            gamsc = lvap[k] * diffu / tcond * rvs_p;
            alphsc = 0.5f * (gamsc / (1.f + gamsc)) * (gamsc / (1.f + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
            alphsc = fmaxf(1.E-9f, alphsc);
            xsat = fminf(-1.E-9f, ssatw);
            t1_evap = 2.f * PI2 * (1.0f - alphsc * xsat + 2.f * alphsc * alphsc * xsat * xsat
                                   - 5.f * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1.f + gamsc);
            lamr = 1. / ilamr;
            if (qv[k] / qvs < 0.95f && rr[k] / rho[k] <= 1.E-8f) {
                prv_rev = rr[k] / rho[k] * odts;
            } else {
                prv_rev = (double)(t1_evap * diffu * (-ssatw)) * N0_r * (double)rvs
                             * ((double)T->t1_qr_ev * d_pow(ilamr, (double)cre[9])
                                + (double)(T->t2_qr_ev * vsc2 * rhof2) * d_pow(lamr + (double)(0.5f * fv_r), -(double)cre[10]));
                rate_max = fminf((rr[k] / rho[k] * odts), (qvs - qv[k]) * odts);
                prv_rev = fmin((double)rate_max, prv_rev / (double)rho[k]);
            }
            pnr_rev = fmin((double)(nr[k] * 0.99f / rho[k] * odts), prv_rev * (double)nr[k] / (double)rr[k]);
            qrten[k] = (float)(qrten[k] - prv_rev);
            qvten[k] = (float)(qvten[k] + prv_rev);
            nrten[k] = (float)(nrten[k] - pnr_rev);
            tten[k] = (float)(tten[k] - (double)(lvap[k] * ocp[k]) * prv_rev * 1);
            rr[k] = fmaxf(R1, (qr1d[k] + dt * qrten[k]) * rho[k]);
            qv[k] = fmaxf(1.E-10f, qv1d[k] + dt * qvten[k]);
            nr[k] = fmaxf(R2, (nr1d[k] + dt * nrten[k]) * rho[k]);
            temp[k] = t1d[k] + dt * tten[k];
            rho[k] = 0.622f * pres[k] / (TH_RR2 * temp[k] * (qv[k] + 0.622f));
        }
        }
    /* ---- :2379-2396 graupel intercept/slope again ---- */
    N0_min = TH_gonv_max;
    for (k = kte; k >= kts; --k) {
        xslw1 = xslw_arr[k];
        ygra1 = 4.31f + d_log10f(fmaxf(5.E-5f, rg[k]));
        zans1 = 3.1f + (100.f / (300.f * xslw1 * ygra1 / (10.f / xslw1 + 1.f + 0.25f * ygra1) + 30.f + 10.f * ygra1));
        N0_exp = d_pow10f(zans1);
        N0_exp = fmax((double)TH_gonv_min, fmin(N0_exp, (double)TH_gonv_max));
        N0_min = fmin(N0_exp, N0_min);
        N0_exp = N0_min;
        lam_exp = d_pow(N0_exp * am_g * cgg[0] / rg[k], (double)T->oge1);
        lamg = lam_exp * d_powf(cgg[2] * T->ogg2 * T->ogg1, T->obmg);
        ilamg[k] = 1. / lamg;
    }    /* ---- :2515-2650 terminal fall speeds and sub-step counts ---- */
    nstep = 0;
    for (n = 0; n < 4; ++n) { onstep[n] = 1.0f; ksed1[n] = 0; }
    for (k = kte + 1; k >= kts; --k) { vtrk[k] = 0.f; vtnrk[k] = 0.f; vtik[k] = 0.f; vtnik[k] = 0.f; vtsk[k] = 0.f; vtgk[k] = 0.f; }
    for (k = kte; k >= kts; --k) {
        vtr = 0.f;
        rhof[k] = sqrtf(TH_rho_not / rho[k]);
        if (rr[k] > R1) {
            lamr = d_powf(am_r * crg[2] * T->org2 * nr[k] / rr[k], T->obmr);
            vtr = (float)((double)(rhof[k] * TH_av_r * crg[5] * T->org3) * d_pow(lamr, (double)cre[2]) * d_pow(lamr + (double)fv_r, -(double)cre[5]));
            vtrk[k] = vtr;
            vtr = (float)((double)(rhof[k] * TH_av_r * crg[6] / crg[11]) * d_pow(lamr, (double)cre[11]) * d_pow(lamr + (double)fv_r, -(double)cre[6]));
            vtnrk[k] = vtr;
        } else { vtrk[k] = vtrk[k + 1]; vtnrk[k] = vtnrk[k + 1]; }
        if (fmaxf(vtrk[k], vtnrk[k]) > 1.E-3f) {
            ksed1[0] = imax(ksed1[0], k);
            delta_tp = dzq[k] / (fmaxf(vtrk[k], vtnrk[k]));
            nstep = imax(nstep, (int)(dt / delta_tp + 1.f));
        }
    }
    if (ksed1[0] == kte) ksed1[0] = kte - 1;
    if (nstep > 0) onstep[0] = 1.f / (float)nstep;

    nstep = 0;
    for (k = kte; k >= kts; --k) {
        vti = 0.f;
        if (ri[k] > R1) {
            lami = d_powf(am_i * cig[1] * T->oig1 * ni[k] / ri[k], T->obmi);
            ilami = 1. / lami;
            vti = (float)((double)(rhof[k] * T->av_i * cig[2] * T->oig2) * ilami);
            vtik[k] = vti;
            vti = (float)((double)(rhof[k] * T->av_i * cig[5] / cig[6]) * ilami);
            vtnik[k] = vti;
        } else { vtik[k] = vtik[k + 1]; vtnik[k] = vtnik[k + 1]; }
        if (vtik[k] > 1.E-3f) {
            ksed1[1] = imax(ksed1[1], k);
            delta_tp = dzq[k] / vtik[k];
            nstep = imax(nstep, (int)(dt / delta_tp + 1.f));
        }
    }
    if (ksed1[1] == kte) ksed1[1] = kte - 1;
    if (nstep > 0) onstep[1] = 1.f / (float)nstep;

    nstep = 0;
    for (k = kte; k >= kts; --k) {
        vts = 0.f;
        if (rs[k] > R1) {
            xDs = smoc[k] / smob[k];
            Mrat = 1.f / xDs;
            ils1 = 1.f / (Mrat * TH_Lam0 + T->fv_s);
            ils2 = 1.f / (Mrat * TH_Lam1 + T->fv_s);
            t1_vts = TH_Kap0 * T->csg[3] * d_powf(ils1, T->cse[3]);
            t2_vts = TH_Kap1 * d_powf(Mrat, TH_mu_s) * T->csg[9] * d_powf(ils2, T->cse[9]);
            ils1 = 1.f / (Mrat * TH_Lam0);
            ils2 = 1.f / (Mrat * TH_Lam1);
            t3_vts = TH_Kap0 * T->csg[0] * d_powf(ils1, T->cse[0]);
            t4_vts = TH_Kap1 * d_powf(Mrat, TH_mu_s) * T->csg[6] * d_powf(ils2, T->cse[6]);
            vts = rhof[k] * T->av_s * (t1_vts + t2_vts) / (t3_vts + t4_vts);
            if (temp[k] > T_0) vtsk[k] = fmaxf(vts * vts_boost[k], vtrk[k]);
            else vtsk[k] = vts * vts_boost[k];
        } else vtsk[k] = vtsk[k + 1];
        if (vtsk[k] > 1.E-3f) {
            ksed1[2] = imax(ksed1[2], k);
            delta_tp = dzq[k] / vtsk[k];
            nstep = imax(nstep, (int)(dt / delta_tp + 1.f));
        }
    }
    if (ksed1[2] == kte) ksed1[2] = kte - 1;
    if (nstep > 0) onstep[2] = 1.f / (float)nstep;

    nstep = 0;
    for (k = kte; k >= kts; --k) {
        vtg = 0.f;
        if (rg[k] > R1) {
            vtg = (float)((double)(rhof[k] * av_g * cgg[5] * T->ogg3) * d_pow(ilamg[k], (double)bv_g));
            if (temp[k] > T_0) vtgk[k] = fmaxf(vtg, vtrk[k]); else vtgk[k] = vtg;
        } else vtgk[k] = vtgk[k + 1];
        if (vtgk[k] > 1.E-3f) {
            ksed1[3] = imax(ksed1[3], k);
            delta_tp = dzq[k] / vtgk[k];
            nstep = imax(nstep, (int)(dt / delta_tp + 1.f));
        }
    }
    if (ksed1[3] == kte) ksed1[3] = kte - 1;
    if (nstep > 0) onstep[3] = 1.f / (float)nstep;
    /* ---- :2660-2770 sedimentation ---- */
    nstep = (int)lroundf(1.f / onstep[0]);
    for (n = 1; n <= nstep; ++n) {
        for (k = kte; k >= kts; --k) { sed_r[k] = vtrk[k] * rr[k]; sed_n[k] = vtnrk[k] * nr[k]; }
        k = kte;
        odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
        qrten[k] = qrten[k] - sed_r[k] * odzq * onstep[0] * orho;
        nrten[k] = nrten[k] - sed_n[k] * odzq * onstep[0] * orho;
        rr[k] = fmaxf(R1, rr[k] - sed_r[k] * odzq * dt * onstep[0]);
        nr[k] = fmaxf(R2, nr[k] - sed_n[k] * odzq * dt * onstep[0]);
        for (k = ksed1[0]; k >= kts; --k) {
            odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
            qrten[k] = qrten[k] + (sed_r[k + 1] - sed_r[k]) * odzq * onstep[0] * orho;
            nrten[k] = nrten[k] + (sed_n[k + 1] - sed_n[k]) * odzq * onstep[0] * orho;
            rr[k] = fmaxf(R1, rr[k] + (sed_r[k + 1] - sed_r[k]) * odzq * dt * onstep[0]);
            nr[k] = fmaxf(R2, nr[k] + (sed_n[k + 1] - sed_n[k]) * odzq * dt * onstep[0]);
        }
        if (rr[kts] > R1 * 10.f) *pptrain = *pptrain + sed_r[kts] * dt * onstep[0];
    }
    nstep = (int)lroundf(1.f / onstep[1]);
    for (n = 1; n <= nstep; ++n) {
        for (k = kte; k >= kts; --k) { sed_r[k] = vtik[k] * ri[k]; sed_n[k] = vtnik[k] * ni[k]; }
        k = kte;
        odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
        qiten[k] = qiten[k] - sed_r[k] * odzq * onstep[1] * orho;
        niten[k] = niten[k] - sed_n[k] * odzq * onstep[1] * orho;
        ri[k] = fmaxf(R1, ri[k] - sed_r[k] * odzq * dt * onstep[1]);
        ni[k] = fmaxf(R2, ni[k] - sed_n[k] * odzq * dt * onstep[1]);
        for (k = ksed1[1]; k >= kts; --k) {
            odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
            qiten[k] = qiten[k] + (sed_r[k + 1] - sed_r[k]) * odzq * onstep[1] * orho;
            niten[k] = niten[k] + (sed_n[k + 1] - sed_n[k]) * odzq * onstep[1] * orho;
            ri[k] = fmaxf(R1, ri[k] + (sed_r[k + 1] - sed_r[k]) * odzq * dt * onstep[1]);
            ni[k] = fmaxf(R2, ni[k] + (sed_n[k + 1] - sed_n[k]) * odzq * dt * onstep[1]);
        }
        if (ri[kts] > R1 * 10.f) *pptice = *pptice + sed_r[kts] * dt * onstep[1];
    }
    nstep = (int)lroundf(1.f / onstep[2]);
    for (n = 1; n <= nstep; ++n) {
        for (k = kte; k >= kts; --k) sed_r[k] = vtsk[k] * rs[k];
        k = kte;
        odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
        qsten[k] = qsten[k] - sed_r[k] * odzq * onstep[2] * orho;
        rs[k] = fmaxf(R1, rs[k] - sed_r[k] * odzq * dt * onstep[2]);
        for (k = ksed1[2]; k >= kts; --k) {
            odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
            qsten[k] = qsten[k] + (sed_r[k + 1] - sed_r[k]) * odzq * onstep[2] * orho;
            rs[k] = fmaxf(R1, rs[k] + (sed_r[k + 1] - sed_r[k]) * odzq * dt * onstep[2]);
        }
        if (rs[kts] > R1 * 10.f) *pptsnow = *pptsnow + sed_r[kts] * dt * onstep[2];
    }
    nstep = (int)lroundf(1.f / onstep[3]);
    for (n = 1; n <= nstep; ++n) {
        for (k = kte; k >= kts; --k) sed_r[k] = vtgk[k] * rg[k];
        k = kte;
        odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
        qgten[k] = qgten[k] - sed_r[k] * odzq * onstep[3] * orho;
        rg[k] = fmaxf(R1, rg[k] - sed_r[k] * odzq * dt * onstep[3]);
        for (k = ksed1[3]; k >= kts; --k) {
            odzq = 1.f / dzq[k]; orho = 1.f / rho[k];
            qgten[k] = qgten[k] + (sed_r[k + 1] - sed_r[k]) * odzq * onstep[3] * orho;
            rg[k] = fmaxf(R1, rg[k] + (sed_r[k + 1] - sed_r[k]) * odzq * dt * onstep[3]);
        }
        if (rg[kts] > R1 * 10.f) *pptgraul = *pptgraul + sed_r[kts] * dt * onstep[3];
    }
    /* ---- :2777-2794 instant melt / freeze ---- */
    for (k = kts; k <= kte; ++k) {
        xri = fmaxf(0.0f, qi1d[k] + qiten[k] * dt);
        if ((temp[k] > T_0) && (xri > 0.0f)) {
            qcten[k] = qcten[k] + xri * odt;
            qiten[k] = qiten[k] - xri * odt;
            niten[k] = -ni1d[k] * odt;
            tten[k] = tten[k] - TH_lfus * ocp[k] * xri * odt * 1;
        }
        xrc = fmaxf(0.0f, qc1d[k] + qcten[k] * dt);
        if ((temp[k] < HGFR) && (xrc > 0.0f)) {
            lfus2 = lsub - lvap[k];
            qiten[k] = qiten[k] + xrc * odt;
            niten[k] = niten[k] + xrc / xm0i * odt;
            qcten[k] = qcten[k] - xrc * odt;
            tten[k] = tten[k] + lfus2 * ocp[k] * xrc * odt * 1;
        }
    }
    /* ---- :2800-2842 apply tendencies ---- */
    for (k = kts; k <= kte; ++k) {
        t1d[k] = t1d[k] + tten[k] * dt;
        qv1d[k] = fmaxf(1.E-10f, qv1d[k] + qvten[k] * dt);
        qc1d[k] = qc1d[k] + qcten[k] * dt;
        if (qc1d[k] <= R1) qc1d[k] = 0.0f;
        qi1d[k] = qi1d[k] + qiten[k] * dt;
        ni1d[k] = fmaxf(R2 / rho[k], ni1d[k] + niten[k] * dt);
        if (qi1d[k] <= R1) { qi1d[k] = 0.0f; ni1d[k] = 0.0f; }
        else {
            lami = d_powf(am_i * cig[1] * T->oig1 * ni1d[k] / qi1d[k], T->obmi);
            ilami = 1. / lami;
            xDi = (float)((double)(bm_i + mu_i + 1.f) * ilami);
            if (xDi < 20.E-6f) lami = cie[1] / 20.E-6f;
            else if (xDi > 300.E-6f) lami = cie[1] / 300.E-6f;
            ni1d[k] = (float)fmin((double)(cig[0] * T->oig2 * qi1d[k] / am_i) * (lami * lami * lami), 250.e3 / (double)rho[k]);
        }
        qr1d[k] = qr1d[k] + qrten[k] * dt;
        nr1d[k] = fmaxf(R2 / rho[k], nr1d[k] + nrten[k] * dt);
        if (qr1d[k] <= R1) { qr1d[k] = 0.0f; nr1d[k] = 0.0f; }
        else {
            lamr = d_powf(am_r * crg[2] * T->org2 * nr1d[k] / qr1d[k], T->obmr);
            mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
            if (mvd_r[k] > 2.5E-3f) mvd_r[k] = 2.5E-3f;
            else if (mvd_r[k] < D0r * 0.75f) mvd_r[k] = D0r * 0.75f;
            lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
            nr1d[k] = (float)((double)(crg[1] * T->org3 * qr1d[k]) * (lamr * lamr * lamr) / (double)am_r);
        }
        qs1d[k] = qs1d[k] + qsten[k] * dt;
        if (qs1d[k] <= R1) qs1d[k] = 0.0f;
        qg1d[k] = qg1d[k] + qgten[k] * dt;
        if (qg1d[k] <= R1) qg1d[k] = 0.0f;
    }
    (void)rgvm; (void)xnc; (void)sump;

}


template <int KMAX>
__global__ void __launch_bounds__(64)
k_thompson(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
           float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
           float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
           double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
           float dt, int i0, int i1, int j0, int k0, int nk)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x;
    const int j = j0 + blockIdx.y;
    if (i > i1) return;
    float qv1d[KMAX], qc1d[KMAX], qi1d[KMAX], qr1d[KMAX], qs1d[KMAX], qg1d[KMAX], ni1d[KMAX], nr1d[KMAX], t1d[KMAX], p1d[KMAX], dz1d[KMAX];
    float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
    const int c0 = d.idx(i, k0, j);
    for (int k = 0; k < nk; ++k) {
        const int c = c0 + k * d.sk;
        t1d[k] = th[c] * pii[c]; p1d[k] = p[c]; dz1d[k] = dz[c]; qv1d[k] = qv[c]; qc1d[k] = qc[c]; qi1d[k] = qi[c];
        qr1d[k] = qr[c]; qs1d[k] = qs[c]; qg1d[k] = qg[c]; ni1d[k] = ni[c]; nr1d[k] = nr[c];
    }
    th_column<KMAX>(T, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, dz1d, &pptrain, &pptsnow, &pptgraul, &pptice, nk, dt);
    // mp_gt_driver :908-912 sums into REAL(4) tile arrays that process_subdomain zeroed, then
    // mp_driver.f90:587-595 adds those to the REAL(8) accumulators
    const int c2 = i + d.nx * j;
    const float rainnc = 0.f + pptrain + pptsnow + pptgraul + pptice;
    const float snownc = 0.f + pptsnow + pptice;
    const float graupelnc = 0.f + pptgraul;
    rain_acc[c2] = rain_acc[c2] + rainnc;
    snow_acc[c2] = snow_acc[c2] + snownc;
    graupel_acc[c2] = graupel_acc[c2] + graupelnc;
    for (int k = 0; k < nk; ++k) {
        const int c = c0 + k * d.sk;
        // :997-1010: the inner re-test reads qv1d again, so the stored value is always 1e-7
        qv[c] = (qv1d[k] < 1.E-7f) ? 1.E-7f : qv1d[k];
        qc[c] = qc1d[k]; qi[c] = qi1d[k]; qr[c] = qr1d[k]; qs[c] = qs1d[k]; qg[c] = qg1d[k];
        ni[c] = ni1d[k]; nr[c] = nr1d[k];
        th[c] = t1d[k] / pii[c];
    }
}

#include "thompson_lane.inc"

// one column per wave (4 columns per 256-thread block), one level per lane; see thompson_lane.inc
__global__ void __launch_bounds__(256, 4)   // 4 waves/SIMD (128 VGPRs, 252 B spill) measured best of 2..8: 7.2/6.0/5.6/6.2/6.3/9.2 ms
k_thompson_lane(Dims d, const ThState *__restrict__ T, float *__restrict__ qv, float *__restrict__ qc, float *__restrict__ qr,
                float *__restrict__ qi, float *__restrict__ qs, float *__restrict__ qg, float *__restrict__ ni, float *__restrict__ nr,
                float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ p, const float *__restrict__ dz,
                double *__restrict__ rain_acc, double *__restrict__ snow_acc, double *__restrict__ graupel_acc,
                float dt, int i0, int i1, int j0, int k0, int nk)
{
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

struct ThTiles { int n, i0[4], i1[4], j0[4], ib0[4], nbx[4], off[5], xcd_run; };

// cpb whole columns per block (aligned to multiples of cpb in i), thread = level*cpb + column (thompson_lane.inc: BlockComm)
#ifndef TH_PACK_MINW
#define TH_PACK_MINW 4
#endif
#ifndef TH_PACK_ATTR
#define TH_PACK_ATTR
#endif
#ifndef TH_PACK_MAXT
#define TH_PACK_MAXT 1024
#endif
__global__ void __launch_bounds__(TH_PACK_MAXT, TH_PACK_MINW) TH_PACK_ATTR      // any block size <= 1024; register budget for 4 waves per SIMD (128 VGPRs)
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
    const int first = (tl.ib0[t] + local % tl.nbx[t]) * cpb;  // first column slot of this block (multiple of cpb)
    BlockComm x(lds_pack, threadIdx.x, blockDim.x, cpb, nk, i0 - first, i1 - first);
    const int j = tl.j0[t] + local / tl.nbx[t];
    const int i = x.active ? first + x.col : max(i0, min(i1, first));
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
// arguments come from the host so that nothing is folded at compile time: the value must be what a level computes at run time
__global__ void k_thompson_constants(ThState *T, float rg, float xslw1) { T->N0_exp_default = th_graupel_N0_exp(rg, xslw1); }
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
    // A/B switches for profiling: ICAR_HIP_THOMPSON=lane (column per lane, scratch arrays) | wave (column per wave)
    const char *mode = getenv("ICAR_HIP_THOMPSON");
    const bool want_lane = mode && !strcmp(mode, "lane"), want_wave = mode && !strcmp(mode, "wave");
    // Packed layout (column_comm.h) unless one column per 64-lane wave fills the lanes as well (52 <= nk <= 64).
    int cpb = 0, nt = 0;
    if (nk >= 2 && !want_wave && !want_lane) {
        const int force = getenv("ICAR_HIP_THOMPSON_CPB") ? atoi(getenv("ICAR_HIP_THOMPSON_CPB")) : 0;   // profiling only
        if (force) { cpb = force; nt = (force * nk + 63) / 64 * 64; if (nt > 1024) { cpb = 0; nt = 0; } }
        else {
            const float u = block_comm_geometry(nk, nt, cpb);
            if (nk <= 64 && u <= nk / 64.0f + 0.02f) { cpb = 0; nt = 0; }
        }
    }
    if (cpb) {
        // all tiles in ONE launch: process_halo's four 1-cell strips are latency-bound when launched one after another
        ThTiles tl; tl.n = nt_; tl.off[0] = 0;
        tl.xcd_run = 64;                 // consecutive column groups (and a few rows of them) per XCD turn
        for (int t = 0; t < nt_; ++t) {
            tl.i0[t] = T4[t][0] - c->ims; tl.i1[t] = T4[t][1] - c->ims; tl.j0[t] = T4[t][2] - c->jms;
            tl.ib0[t] = tl.i0[t] / cpb; tl.nbx[t] = tl.i1[t] / cpb - tl.ib0[t] + 1;
            tl.off[t + 1] = tl.off[t] + tl.nbx[t] * (T4[t][3] - T4[t][2] + 1);
        }
        hipLaunchKernelGGL(k_thompson_pack, dim3(tl.off[nt_]), dim3(nt), BlockComm::lds_bytes(nt, cpb), c->stream, c->d, T,
                           qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, dt, tl, kts - c->kms, nk, cpb);
        HIPCHK(hipGetLastError());
        return 0;
    }
    for (int t = 0; t < nt_; ++t) {
        const int its = T4[t][0], i_end = T4[t][1], jts = T4[t][2], j_end = T4[t][3];
        dim3 g((i_end - its + 1 + 63) / 64, j_end - jts + 1), b(64);
#define LAUNCH(K) hipLaunchKernelGGL((k_thompson<K>), g, b, 0, c->stream, c->d, T, qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga, \
                                     dt, its - c->ims, i_end - c->ims, jts - c->jms, kts - c->kms, nk)
        if (nk <= 64 && !want_lane) {
            dim3 gl((i_end - its + 1 + 3) / 4, j_end - jts + 1), bl(256);
            hipLaunchKernelGGL(k_thompson_lane, gl, bl, 0, c->stream, c->d, T, qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, pa, sa, ga,
                               dt, its - c->ims, i_end - c->ims, jts - c->jms, kts - c->kms, nk);
        }
        else if (nk <= 40) LAUNCH(40);
        else if (nk <= 64) LAUNCH(64);
        else if (nk <= 96) LAUNCH(96);
        else { icar_set_error("thompson: this many levels are not supported by this build"); return 1; }
#undef LAUNCH
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
