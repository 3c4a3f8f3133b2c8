/* oracle/thompson_column.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * Column physics of the Thompson scheme: src/physics/mp_thompson.f90:1057-2844 (mp_thompson)
 * and the driver mp_gt_driver :772-1044.  REAL -> float, DOUBLE PRECISION -> double; expression
 * order and mixed-precision promotions follow the reference statement by statement. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "thompson_oracle.h"

#define KMAX 256

/* Transcendental mode (see icar_oracle.c: orc_set_math_mode).  Mode 0 = libm float functions exactly as
 * the flang-compiled reference calls them (bit-identical to oracle/_ref).  Mode 1 = the same functions
 * evaluated in FP64 and rounded once, which is how the HIP kernel defines them. */
extern int g_math_mode;
/* Optional trace of every transcendental call of the column physics (diagnostics for a device / oracle difference: which libm
 * call does the device evaluate differently?).  orc_thompson_trace(1) starts a trace (run ONE column, one thread), the entries are
 * (op, x, y, result) with op = the op codes of icar_probe_math (tests/support/th_probe.hip): 0 log, 1 exp, 2 pow (double); 3 powf, 4 expf,
 * 6 log10f; 10 = log10 (double, no device probe). */
#define TH_TRACE_MAX 65536
static int g_tr_on = 0, g_tr_n = 0;
static double g_tr[4 * TH_TRACE_MAX];
void orc_thompson_trace(int on) { g_tr_on = on; if (on) g_tr_n = 0; }
int orc_thompson_trace_read(double *out, int cap)
{
    const int n = g_tr_n < cap ? g_tr_n : cap;
    memcpy(out, g_tr, sizeof(double) * 4 * (size_t)n);
    return g_tr_n;
}
static inline double tr_rec(int op, double x, double y, double r)
{
    if (g_tr_on && g_tr_n < TH_TRACE_MAX) { double *e = g_tr + 4 * g_tr_n++; e[0] = op; e[1] = x; e[2] = y; e[3] = r; }
    return r;
}
static inline float M_powf(float x, float y) { return (float)tr_rec(3, x, y, g_math_mode ? (float)pow((double)x, (double)y) : powf(x, y)); }
static inline float M_expf(float x) { return (float)tr_rec(4, x, 0, g_math_mode ? (float)exp((double)x) : expf(x)); }
static inline float M_log10f(float x) { return (float)tr_rec(6, x, 0, g_math_mode ? (float)log10((double)x) : log10f(x)); }
static inline double T_pow(double x, double y) { return tr_rec(2, x, y, pow(x, y)); }
static inline double T_log(double x) { return tr_rec(0, x, 0, log(x)); }
static inline double T_exp(double x) { return tr_rec(1, x, 0, exp(x)); }
static inline double T_log10(double x) { return tr_rec(10, x, 0, log10(x)); }
#define pow(x, y) T_pow(x, y)
#define log(x) T_log(x)
#define exp(x) T_exp(x)
#define log10(x) T_log10(x)
#define IDX3(i,k,j) ((size_t)(i) + (size_t)nx*((size_t)(k) + (size_t)nz*(size_t)(j)))

/* 10.**nn with an INTEGER exponent: flang calls __powisf2 (repeated squaring) */
static inline float powi10f(int b)
{
    const int recip = b < 0;
    float a = 10.0f, r = 1.0f;
    if (recip) b = -b;
    while (1) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return recip ? 1.0f / r : r;
}

/* decade-table index: :1562-1574 and siblings (REAL argument) */
static inline int dec_index_f(float r, int n2)
{
    const int nic = (int)lroundf(M_log10f(r));
    int n = nic - 1;
    for (int nn = nic - 1; nn <= nic + 1; ++nn) {
        n = nn;
        if ((r / powi10f(nn)) >= 1.0f && (r / powi10f(nn)) < 10.0f) break;
    }
    return (int)(r / powi10f(n)) + 10 * (n - n2) - (n - n2);
}

/* same with a DOUBLE PRECISION argument (:1620-1627) */
static inline int dec_index_d(double r, int n2)
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
static inline float cube_f(float x) { return x * x * x; }

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

static float rslf(float P, float T)
{   /* :3776-3805 */
    const float C0 = .611583699E03f, C1 = .444606896E02f, C2 = .143177157E01f, C3 = .264224321E-1f, C4 = .299291081E-3f,
                C5 = .203154182E-5f, C6 = .702620698E-8f, C7 = .379534310E-11f, C8 = -.321582393E-13f;
    const float X = fmaxf(-80.f, T - 273.16f);
    const float ESL = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    return .622f * ESL / (P - ESL);
}

static float rsif(float P, float T)
{   /* :3810-3835 */
    const float C0 = .609868993E03f, C1 = .499320233E02f, C2 = .184672631E01f, C3 = .402737184E-1f, C4 = .565392987E-3f,
                C5 = .521693933E-5f, C6 = .307839583E-7f, C7 = .105785160E-9f, C8 = .161444444E-12f;
    const float X = fmaxf(-80.f, T - 273.16f);
    const float ESI = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    return .622f * ESI / (P - ESI);
}

/* Field et al. (2005) moment polynomial in REAL arithmetic (:1379-1449); b = moment order */
static inline float snow_poly_f(const float *s, float tc0, float b)
{
    return s[0] + s[1] * tc0 + s[2] * b + s[3] * tc0 * b + s[4] * tc0 * tc0 + s[5] * b * b + s[6] * tc0 * tc0 * b
         + s[7] * tc0 * b * b + s[8] * tc0 * tc0 * tc0 + s[9] * b * b * b;
}

const float *th_sa_ptr(void);
const float *th_sb_ptr(void);

#define T4S(tab) (TH.tab[(idx_s - 1) + NTB_S * ((idx_t - 1) + NTB_T * ((size_t)(idx_r1 - 1) + NTB_R1 * (idx_r - 1)))])
#define T4G(tab) (TH.tab[(idx_g1 - 1) + NTB_G1 * ((idx_g - 1) + NTB_G * ((size_t)(idx_r1 - 1) + NTB_R1 * (idx_r - 1)))])
#define T3R(tab) (TH.tab[(idx_r - 1) + NTB_R * ((idx_r1 - 1) + NTB_R1 * (size_t)(idx_tc - 1))])
#define T2C(tab) (TH.tab[(idx_c - 1) + NTB_C * (size_t)(idx_tc - 1)])
#define T2I(tab) (TH.tab[(idx_i - 1) + NTB_I * (size_t)(idx_i1 - 1)])

/* mp_thompson :1057-2844.  Arrays are 0-based k = 0..nz-1 (kts..kte). */
void th_column(float *qv1d, float *qc1d, float *qi1d, float *qr1d, float *qs1d, float *qg1d, float *ni1d, float *nr1d,
               float *t1d, float *p1d, const float *dzq, float *pptrain, float *pptsnow, float *pptgraul, float *pptice,
               int nz, float dt)
{
    const float *sa = th_sa_ptr(), *sb = th_sb_ptr();
    const float R1 = TH_R1, R2 = TH_R2, eps = TH_eps, T_0 = TH_T_0, PI2 = TH_PI2;
    const float am_r = TH_am_r, am_i = TH_am_i, bm_r = TH_bm_r, bm_i = TH_bm_i, bm_g = TH_bm_g, mu_i = TH_mu_i, mu_g = TH_mu_g;
    const float D0r = TH_D0r, D0c = TH_D0c, D0s = TH_D0s, D0g = TH_D0g, fv_r = TH_fv_r, lsub = TH_lsub, lvap0 = TH_lvap0;
    const float oRv = TH_oRv, olfus = TH_olfus, xm0i = TH_xm0i, C_cube = TH_C_cube, HGFR = TH_HGFR, rho_w = TH_rho_w;
    const float mu_r = TH.mu_r, mu_c = TH.mu_c, Nt_c = TH.Nt_c, am_g = TH.am_g, av_g = TH.av_g, bv_g = TH.bv_g;
    const float *cce = TH.cce, *ccg = TH.ccg, *cie = TH.cie, *cig = TH.cig, *cre = TH.cre, *crg = TH.crg, *cse = TH.cse,
                *cge = TH.cge, *cgg = TH.cgg;
    (void)cce; (void)cse;
    const int kts = 0, kte = nz - 1;

    static __thread float tten[KMAX], qvten[KMAX], qcten[KMAX], qiten[KMAX], qrten[KMAX], qsten[KMAX], qgten[KMAX], niten[KMAX], nrten[KMAX];
    static __thread double prw_vcd[KMAX];
    static __thread double prr_wau[KMAX], prr_rcw[KMAX], prr_rcs[KMAX], prr_rcg[KMAX], prr_sml[KMAX], prr_gml[KMAX], prr_rci[KMAX], prv_rev[KMAX],
        pnr_wau[KMAX], pnr_rcs[KMAX], pnr_rcg[KMAX], pnr_rci[KMAX], pnr_sml[KMAX], pnr_gml[KMAX], pnr_rev[KMAX], pnr_rcr[KMAX], pnr_rfz[KMAX];
    static __thread double pri_inu[KMAX], pni_inu[KMAX], pri_ihm[KMAX], pni_ihm[KMAX], pri_wfz[KMAX], pni_wfz[KMAX], pri_rfz[KMAX], pni_rfz[KMAX],
        pri_ide[KMAX], pni_ide[KMAX], pri_rci[KMAX], pni_rci[KMAX], pni_sci[KMAX], pni_iau[KMAX];
    static __thread double prs_iau[KMAX], prs_sci[KMAX], prs_rcs[KMAX], prs_scw[KMAX], prs_sde[KMAX], prs_ihm[KMAX], prs_ide[KMAX];
    static __thread double prg_scw[KMAX], prg_rfz[KMAX], prg_gde[KMAX], prg_gcw[KMAX], prg_rci[KMAX], prg_rcs[KMAX], prg_rcg[KMAX], prg_ihm[KMAX];
    static __thread float temp[KMAX], pres[KMAX], qv[KMAX], rc[KMAX], ri[KMAX], rr[KMAX], rs[KMAX], rg[KMAX], ni[KMAX], nr[KMAX];
    static __thread float rho[KMAX], rhof[KMAX], rhof2[KMAX], qvs[KMAX], qvsi[KMAX], delQvs[KMAX], satw[KMAX], sati[KMAX], ssatw[KMAX], ssati[KMAX];
    static __thread float diffu[KMAX], visco[KMAX], vsc2[KMAX], tcond[KMAX], lvap[KMAX], ocp[KMAX], lvt2[KMAX];
    static __thread double ilamr[KMAX], ilamg[KMAX], N0_r[KMAX], N0_g[KMAX];
    static __thread float mvd_r[KMAX], mvd_c[KMAX];
    static __thread float smob[KMAX], smo2[KMAX], smo1[KMAX], smo0[KMAX], smoc[KMAX], smod[KMAX], smoe[KMAX], smof[KMAX];
    static __thread float sed_r[KMAX], sed_s[KMAX], sed_g[KMAX], sed_i[KMAX], sed_n[KMAX];
    static __thread float vtik[KMAX + 1], vtnik[KMAX + 1], vtrk[KMAX + 1], vtnrk[KMAX + 1], vtsk[KMAX + 1], vtgk[KMAX + 1];
    static __thread float vts_boost[KMAX];
    static __thread int L_qc[KMAX], L_qi[KMAX], L_qr[KMAX], L_qs[KMAX], L_qg[KMAX];
    (void)smod; (void)satw; (void)sati;

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

#define Z(a) memset(a, 0, sizeof(a[0]) * nz)
    Z(tten); Z(qvten); Z(qcten); Z(qiten); Z(qrten); Z(qsten); Z(qgten); Z(niten); Z(nrten); Z(prw_vcd);
    Z(prv_rev); Z(prr_wau); Z(prr_rcw); Z(prr_rcs); Z(prr_rcg); Z(prr_sml); Z(prr_gml); Z(prr_rci); Z(pnr_wau); Z(pnr_rcs);
    Z(pnr_rcg); Z(pnr_rci); Z(pnr_sml); Z(pnr_gml); Z(pnr_rev); Z(pnr_rcr); Z(pnr_rfz);
    Z(pri_inu); Z(pni_inu); Z(pri_ihm); Z(pni_ihm); Z(pri_wfz); Z(pni_wfz); Z(pri_rfz); Z(pni_rfz); Z(pri_ide); Z(pni_ide);
    Z(pri_rci); Z(pni_rci); Z(pni_sci); Z(pni_iau);
    Z(prs_iau); Z(prs_sci); Z(prs_rcs); Z(prs_scw); Z(prs_sde); Z(prs_ihm); Z(prs_ide);
    Z(prg_scw); Z(prg_rfz); Z(prg_gde); Z(prg_gcw); Z(prg_rci); Z(prg_rcs); Z(prg_rcg); Z(prg_ihm);
    Z(smob); Z(smo2); Z(smo1); Z(smo0); Z(smoc); Z(smoe); Z(smof);   /* never read uninitialised below; zero for determinism */
#undef Z

    /* ---- :1240-1319 column -> local arrays ---- */
    for (k = kts; k <= kte; ++k) {
        temp[k] = t1d[k];
        qv[k] = fmaxf(1.E-10f, qv1d[k]);
        pres[k] = p1d[k];
        rho[k] = 0.622f * pres[k] / (TH_RR2 * temp[k] * (qv[k] + 0.622f));
        if (qc1d[k] > R1) { no_micro = 0; rc[k] = qc1d[k] * rho[k]; L_qc[k] = 1; }
        else { qc1d[k] = 0.0f; rc[k] = R1; L_qc[k] = 0; }
        if (qi1d[k] > R1) {
            no_micro = 0;
            ri[k] = qi1d[k] * rho[k];
            ni[k] = fmaxf(R2, ni1d[k] * rho[k]);
            L_qi[k] = 1;
            lami = M_powf(am_i * cig[1] * TH.oig1 * ni[k] / ri[k], TH.obmi);
            ilami = 1. / lami;
            xDi = (float)((double)(bm_i + mu_i + 1.f) * ilami);
            if (xDi < 20.E-6f) {
                lami = cie[1] / 20.E-6f;
                ni[k] = (float)fmin(250.e3, (double)(cig[0] * TH.oig2 * ri[k] / am_i) * (lami * lami * lami));
            } else if (xDi > 300.E-6f) {
                lami = cie[1] / 300.E-6f;
                ni[k] = (float)((double)(cig[0] * TH.oig2 * ri[k] / am_i) * (lami * lami * lami));
            }
        } else { qi1d[k] = 0.0f; ni1d[k] = 0.0f; ri[k] = R1; ni[k] = R2; L_qi[k] = 0; }

        mvd_r[k] = 0.0f;
        if (qr1d[k] > R1) {
            no_micro = 0;
            rr[k] = qr1d[k] * rho[k];
            nr[k] = fmaxf(R2, nr1d[k] * rho[k]);
            L_qr[k] = 1;
            lamr = M_powf(am_r * crg[2] * TH.org2 * nr[k] / rr[k], TH.obmr);
            mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
            if (mvd_r[k] > 2.5E-3f) {
                mvd_r[k] = 2.5E-3f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * TH.org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            } else if (mvd_r[k] < D0r * 0.75f) {
                mvd_r[k] = D0r * 0.75f;
                lamr = (3.0f + mu_r + 0.672f) / mvd_r[k];
                nr[k] = (float)((double)(crg[1] * TH.org3 * rr[k]) * (lamr * lamr * lamr) / (double)am_r);
            }
        } else { qr1d[k] = 0.0f; nr1d[k] = 0.0f; rr[k] = R1; nr[k] = R2; L_qr[k] = 0; }
        if (qs1d[k] > R1) { no_micro = 0; rs[k] = qs1d[k] * rho[k]; L_qs[k] = 1; }
        else { qs1d[k] = 0.0f; rs[k] = R1; L_qs[k] = 0; }
        if (qg1d[k] > R1) { no_micro = 0; rg[k] = qg1d[k] * rho[k]; L_qg[k] = 1; }
        else { qg1d[k] = 0.0f; rg[k] = R1; L_qg[k] = 0; }
    }

    /* ---- :1328-1356 thermodynamics ---- */
    for (k = kts; k <= kte; ++k) {
        tempc = temp[k] - 273.15f;
        rhof[k] = sqrtf(TH_rho_not / rho[k]);
        rhof2[k] = sqrtf(rhof[k]);
        qvs[k] = rslf(pres[k], temp[k]);
        delQvs[k] = fmaxf(0.0f, rslf(pres[k], 273.15f) - qv[k]);
        if (tempc <= 0.0f) qvsi[k] = rsif(pres[k], temp[k]); else qvsi[k] = qvs[k];
        satw[k] = qv[k] / qvs[k];
        sati[k] = qv[k] / qvsi[k];
        ssatw[k] = satw[k] - 1.f;
        ssati[k] = sati[k] - 1.f;
        if (fabsf(ssatw[k]) < eps) ssatw[k] = 0.0f;
        if (fabsf(ssati[k]) < eps) ssati[k] = 0.0f;
        if (no_micro && ssati[k] > 0.0f) no_micro = 0;
        diffu[k] = 2.11E-5f * M_powf(temp[k] / 273.15f, 1.94f) * (101325.f / pres[k]);
        if (tempc >= 0.0f) visco[k] = (1.718f + 0.0049f * tempc) * 1.0E-5f;
        else visco[k] = (1.718f + 0.0049f * tempc - 1.2E-5f * tempc * tempc) * 1.0E-5f;
        ocp[k] = 1.f / (TH_Cp2 * (1.f + 0.887f * qv[k]));
        vsc2[k] = sqrtf(rho[k] / visco[k]);
        lvap[k] = lvap0 + (2106.0f - 4218.0f) * tempc;
        tcond[k] = (5.69f + 0.0168f * tempc) * 1.0E-5f * 418.936f;
    }

    if (no_micro) return;     /* :1363 */

    /* ---- :1369-1451 snow moments ---- */
    for (k = kts; k <= kte; ++k) {
        if (!L_qs[k]) continue;
        tc0 = fminf(-0.1f, temp[k] - 273.15f);
        smob[k] = rs[k] * TH.oams;
        if (TH_bm_s > (2.0f - 1.e-3f) && TH_bm_s < (2.0f + 1.e-3f)) smo2[k] = smob[k];
        else {
            loga_ = snow_poly_f(sa, tc0, TH_bm_s); a_ = M_powf(10.0f, loga_); b_ = snow_poly_f(sb, tc0, TH_bm_s);
            smo2[k] = M_powf(smob[k] / a_, 1.f / b_);
        }
        loga_ = sa[0] + sa[1] * tc0 + sa[4] * tc0 * tc0 + sa[8] * tc0 * tc0 * tc0;
        a_ = M_powf(10.0f, loga_);
        b_ = sb[0] + sb[1] * tc0 + sb[4] * tc0 * tc0 + sb[8] * tc0 * tc0 * tc0;
        smo0[k] = a_ * M_powf(smo2[k], b_);
        loga_ = sa[0] + sa[1] * tc0 + sa[2] + sa[3] * tc0 + sa[4] * tc0 * tc0 + sa[5] + sa[6] * tc0 * tc0 + sa[7] * tc0
              + sa[8] * tc0 * tc0 * tc0 + sa[9];
        a_ = M_powf(10.0f, loga_);
        b_ = sb[0] + sb[1] * tc0 + sb[2] + sb[3] * tc0 + sb[4] * tc0 * tc0 + sb[5] + sb[6] * tc0 * tc0 + sb[7] * tc0
           + sb[8] * tc0 * tc0 * tc0 + sb[9];
        smo1[k] = a_ * M_powf(smo2[k], b_);
        loga_ = snow_poly_f(sa, tc0, TH.cse[0]); a_ = M_powf(10.0f, loga_); b_ = snow_poly_f(sb, tc0, TH.cse[0]);
        smoc[k] = a_ * M_powf(smo2[k], b_);
        loga_ = snow_poly_f(sa, tc0, TH.cse[12]); a_ = M_powf(10.0f, loga_); b_ = snow_poly_f(sb, tc0, TH.cse[12]);
        smoe[k] = a_ * M_powf(smo2[k], b_);
        loga_ = snow_poly_f(sa, tc0, TH.cse[15]); a_ = M_powf(10.0f, loga_); b_ = snow_poly_f(sb, tc0, TH.cse[15]);
        smof[k] = a_ * M_powf(smo2[k], b_);
    }

    /* ---- :1456-1482 graupel intercept/slope, top-down running minimum ---- */
    N0_min = TH_gonv_max;
    for (k = kte; k >= kts; --k) {
        if (temp[k] < 270.65f && L_qr[k] && mvd_r[k] > 100.E-6f) xslw1 = 4.01f + M_log10f(mvd_r[k]);
        else xslw1 = 0.01f;
        ygra1 = 4.31f + M_log10f(fmaxf(5.E-5f, rg[k]));
        zans1 = 3.1f + (100.f / (300.f * xslw1 * ygra1 / (10.f / xslw1 + 1.f + 0.25f * ygra1) + 30.f + 10.f * ygra1));
        N0_exp = M_powf(10.f, zans1);
        N0_exp = fmax((double)TH_gonv_min, fmin(N0_exp, (double)TH_gonv_max));
        N0_min = fmin(N0_exp, N0_min);
        N0_exp = N0_min;
        lam_exp = pow(N0_exp * am_g * cgg[0] / rg[k], (double)TH.oge1);
        lamg = lam_exp * M_powf(cgg[2] * TH.ogg2 * TH.ogg1, TH.obmg);
        ilamg[k] = 1. / lamg;
        N0_g[k] = N0_exp / (cgg[1] * lam_exp) * pow(lamg, (double)cge[1]);
    }

    /* ---- :1489-1494 rain intercept/slope ---- */
    for (k = kte; k >= kts; --k) {
        lamr = M_powf(am_r * crg[2] * TH.org2 * nr[k] / rr[k], TH.obmr);
        ilamr[k] = 1. / lamr;
        mvd_r[k] = (float)((double)(3.0f + mu_r + 0.672f) / lamr);
        N0_r[k] = (double)(nr[k] * TH.org2) * pow(lamr, (double)cre[1]);
    }

    /* ---- :1500-1544 warm rain ---- */
    for (k = kts; k <= kte; ++k) {
        if (L_qr[k] && mvd_r[k] > D0r) {
            Ef_rr = 2.0f - M_expf(2300.0f * (mvd_r[k] - 1600.0E-6f));
            pnr_rcr[k] = Ef_rr * 4.f * nr[k] * rr[k];
        }
        mvd_c[k] = D0c;
        if (!L_qc[k]) continue;
        xDc = fmaxf(D0c * 1.E6f, (M_powf(rc[k] / (am_r * Nt_c), TH.obmr)) * 1.E6f);
        lamc = M_powf(Nt_c * am_r * ccg[1] * TH.ocg1 / rc[k], TH.obmr);
        mvd_c[k] = (float)((double)(3.0f + mu_c + 0.672f) / lamc);
        if (rc[k] > 0.01e-3f) {
            Dc_g = (float)(((double)M_powf(ccg[2] * TH.ocg2, TH.obmr) / lamc) * (double)1.E6f);
            Dc_b = M_powf(xDc * xDc * xDc * Dc_g * Dc_g * Dc_g - xDc * xDc * xDc * xDc * xDc * xDc, 1.f / 6.f);
            zeta1 = 0.5f * ((6.25E-6f * xDc * Dc_b * Dc_b * Dc_b - 0.4f) + fabsf(6.25E-6f * xDc * Dc_b * Dc_b * Dc_b - 0.4f));
            zeta = 0.027f * rc[k] * zeta1;
            taud = 0.5f * ((0.5f * Dc_b - 7.5f) + fabsf(0.5f * Dc_b - 7.5f)) + R1;
            tau = 3.72f / (rc[k] * taud);
            prr_wau[k] = zeta / tau;
            prr_wau[k] = fmin((double)(rc[k] * odts), prr_wau[k]);
            pnr_wau[k] = prr_wau[k] / (double)(am_r * mu_c * D0r * D0r * D0r);
        }
        if (L_qr[k] && mvd_r[k] > D0r && mvd_c[k] > D0c) {
            lamr = 1. / ilamr[k];
            idx = 1 + (int)(NBINS * log((double)mvd_r[k] / TH.Dr[0]) / log(TH.Dr[NBINS - 1] / TH.Dr[0]));
            idx = imin(idx, NBINS);
            int ic = (int)(mvd_c[k] * 1.E6f);
            ic = imax(1, imin(ic, NBINS));          /* the reference does not bound this index */
            Ef_rw = (float)TH.t_Efrw[(idx - 1) + NBINS * (ic - 1)];
            prr_rcw[k] = (double)(rhof[k] * TH.t1_qr_qc * Ef_rw * rc[k]) * N0_r[k] * pow(lamr + (double)fv_r, -(double)cre[8]);
            prr_rcw[k] = fmin((double)(rc[k] * odts), prr_rcw[k]);
        }
    }

    /* ---- :1550-2009 frozen-species process terms ---- */
    for (k = kts; k <= kte; ++k) {
        vts_boost[k] = 1.5f;
        tempc = temp[k] - 273.15f;
        idx_tc = imax(1, imin((int)lroundf(-tempc), 45));
        idx_t = (int)((tempc - 2.5f) / 5.f) - 1;
        idx_t = imax(1, -idx_t);
        idx_t = imin(idx_t, NTB_T);

        if (rc[k] > TH.r_c[0]) { idx_c = dec_index_f(rc[k], TH.nic2); idx_c = imax(1, imin(idx_c, NTB_C)); } else idx_c = 1;
        if (ri[k] > TH.r_i[0]) { idx_i = dec_index_f(ri[k], TH.nii2); idx_i = imax(1, imin(idx_i, NTB_I)); } else idx_i = 1;
        if (ni[k] > TH.Nt_i[0]) { idx_i1 = dec_index_f(ni[k], TH.nii3); idx_i1 = imax(1, imin(idx_i1, NTB_I1)); } else idx_i1 = 1;
        if (rr[k] > TH.r_r[0]) {
            idx_r = dec_index_f(rr[k], TH.nir2); idx_r = imax(1, imin(idx_r, NTB_R));
            lamr = 1. / ilamr[k];
            lam_exp = lamr * cube_f(crg[2] * TH.org2 * TH.org1);
            N0_exp = (double)(TH.org1 * rr[k] / am_r) * pow(lam_exp, (double)cre[0]);
            idx_r1 = dec_index_d(N0_exp, TH.nir3); idx_r1 = imax(1, imin(idx_r1, NTB_R1));
        } else { idx_r = 1; idx_r1 = NTB_R1; }
        if (rs[k] > TH.r_s[0]) { idx_s = dec_index_f(rs[k], TH.nis2); idx_s = imax(1, imin(idx_s, NTB_S)); } else idx_s = 1;
        if (rg[k] > TH.r_g[0]) {
            idx_g = dec_index_f(rg[k], TH.nig2); idx_g = imax(1, imin(idx_g, NTB_G));
            lamg = 1. / ilamg[k];
            lam_exp = lamg * cube_f(cgg[2] * TH.ogg2 * TH.ogg1);
            N0_exp = (double)(TH.ogg1 * rg[k] / am_g) * pow(lam_exp, (double)cge[0]);
            idx_g1 = dec_index_d(N0_exp, TH.nig3); idx_g1 = imax(1, imin(idx_g1, NTB_G1));
        } else { idx_g = 1; idx_g1 = NTB_G1; }

        /* deposition/sublimation prefactor :1679-1695 */
        otemp = 1.f / temp[k];
        rvs = rho[k] * qvsi[k];
        rvs_p = rvs * otemp * (lsub * otemp * oRv - 1.f);
        rvs_pp = rvs * (otemp * (lsub * otemp * oRv - 1.f) * otemp * (lsub * otemp * oRv - 1.f)
                        + (-2.f * lsub * otemp * otemp * otemp * oRv) + otemp * otemp);
        gamsc = lsub * diffu[k] / tcond[k] * rvs_p;
        alphsc = 0.5f * (gamsc / (1.f + gamsc)) * (gamsc / (1.f + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
        alphsc = fmaxf(1.E-9f, alphsc);
        xsat = ssati[k];
        if (fabsf(xsat) < 1.E-9f) xsat = 0.f;
        t1_subl = 4.f * PI2 * (1.0f - alphsc * xsat + 2.f * alphsc * alphsc * xsat * xsat
                               - 5.f * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1.f + gamsc);

        /* snow / graupel collecting cloud water :1698-1725 */
        if (L_qc[k] && mvd_c[k] > D0c) {
            xDs = 0.0f;
            if (L_qs[k]) xDs = smoc[k] / smob[k];
            if (xDs > D0s) {
                idx = 1 + (int)(NBINS * log((double)xDs / TH.Ds[0]) / log(TH.Ds[NBINS - 1] / TH.Ds[0]));
                idx = imin(idx, NBINS);
                int ic = (int)(mvd_c[k] * 1.E6f); ic = imax(1, imin(ic, NBINS));
                Ef_sw = (float)TH.t_Efsw[(idx - 1) + NBINS * (ic - 1)];
                prs_scw[k] = rhof[k] * TH.t1_qs_qc * Ef_sw * rc[k] * smoe[k];
            }
            if (rg[k] >= TH.r_g[0] && mvd_c[k] > D0c) {
                xDg = (float)((double)(bm_g + mu_g + 1.f) * ilamg[k]);
                vtg = (float)((double)(rhof[k] * av_g * cgg[5] * TH.ogg3) * pow(ilamg[k], (double)bv_g));
                stoke_g = mvd_c[k] * mvd_c[k] * vtg * rho_w / (9.f * visco[k] * xDg);
                if (xDg > D0g) {
                    if (stoke_g >= 0.4f && stoke_g <= 10.f) Ef_gw = 0.55f * M_log10f(2.51f * stoke_g);
                    else if (stoke_g < 0.4f) Ef_gw = 0.0f;
                    else if (stoke_g > 10.f) Ef_gw = 0.77f;
                    prg_gcw[k] = (double)(rhof[k] * TH.t1_qg_qc * Ef_gw * rc[k]) * N0_g[k] * pow(ilamg[k], (double)cge[8]);
                }
            }
        }

        /* rain collecting snow / graupel :1730-1783 */
        if (rr[k] >= TH.r_r[0]) {
            if (rs[k] >= TH.r_s[0]) {
                if (temp[k] < T_0) {
                    prr_rcs[k] = -(T4S(tmr_racs2) + T4S(tcr_sacr2) + T4S(tmr_racs1) + T4S(tcr_sacr1));
                    prs_rcs[k] = T4S(tmr_racs2) + T4S(tcr_sacr2) - T4S(tcs_racs1) - T4S(tms_sacr1);
                    prg_rcs[k] = T4S(tmr_racs1) + T4S(tcr_sacr1) + T4S(tcs_racs1) + T4S(tms_sacr1);
                    prr_rcs[k] = fmax((double)(-rr[k] * odts), prr_rcs[k]);
                    prs_rcs[k] = fmax((double)(-rs[k] * odts), prs_rcs[k]);
                    prg_rcs[k] = fmin((double)((rr[k] + rs[k]) * odts), prg_rcs[k]);
                    pnr_rcs[k] = T4S(tnr_racs1) + T4S(tnr_racs2) + T4S(tnr_sacr1) + T4S(tnr_sacr2);
                } else {
                    prs_rcs[k] = -T4S(tcs_racs1) - T4S(tms_sacr1) + T4S(tmr_racs2) + T4S(tcr_sacr2);
                    prs_rcs[k] = fmax((double)(-rs[k] * odts), prs_rcs[k]);
                    prr_rcs[k] = -prs_rcs[k];
                    pnr_rcs[k] = T4S(tnr_racs2) + T4S(tnr_sacr2);
                }
                pnr_rcs[k] = fmin((double)(nr[k] * odts), pnr_rcs[k]);
            }
            if (rg[k] >= TH.r_g[0]) {
                if (temp[k] < T_0) {
                    prg_rcg[k] = T4G(tmr_racg) + T4G(tcr_gacr);
                    prg_rcg[k] = fmin((double)(rr[k] * odts), prg_rcg[k]);
                    prr_rcg[k] = -prg_rcg[k];
                    pnr_rcg[k] = T4G(tnr_racg) + T4G(tnr_gacr);
                    pnr_rcg[k] = fmin((double)(nr[k] * odts), pnr_rcg[k]);
                } else {
                    prr_rcg[k] = T4G(tcg_racg);
                    prr_rcg[k] = fmin((double)(rg[k] * odts), prr_rcg[k]);
                    prg_rcg[k] = -prr_rcg[k];
                }
            }
        }

        if (temp[k] < T_0) {      /* :1789-1949 sub-zero processes */
            vts_boost[k] = 1.0f;
            rate_max = (qv[k] - qvsi[k]) * rho[k] * odts * 0.999f;
            if (rr[k] > TH.r_r[0]) {
                prg_rfz[k] = T3R(tpg_qrfz) * odts;
                pri_rfz[k] = T3R(tpi_qrfz) * odts;
                pni_rfz[k] = T3R(tni_qrfz) * odts;
                pnr_rfz[k] = T3R(tnr_qrfz) * odts;
                pnr_rfz[k] = fmin((double)(nr[k] * odts), pnr_rfz[k]);
            } else if (rr[k] > R1 && temp[k] < HGFR) {
                pri_rfz[k] = rr[k] * odts;
                pnr_rfz[k] = nr[k] * odts;
                pni_rfz[k] = pnr_rfz[k];
            }
            if (rc[k] > TH.r_c[0]) {
                pri_wfz[k] = T2C(tpi_qcfz) * odts;
                pri_wfz[k] = fmin((double)(rc[k] * odts), pri_wfz[k]);
                pni_wfz[k] = T2C(tni_qcfz) * odts;
                pni_wfz[k] = fmin(fmin((double)(Nt_c * odts), pri_wfz[k] / (double)(2.f * xm0i)), pni_wfz[k]);
            } else if (rc[k] > R1 && temp[k] < HGFR) {
                pri_wfz[k] = rc[k] * odts;
                pni_wfz[k] = fmin(fmin((double)(Nt_c * odts), pri_wfz[k] / (double)(2.f * xm0i)), pni_wfz[k]);
            }
            if ((ssati[k] >= 0.25f) || (ssatw[k] > eps && temp[k] < 261.15f)) {
                xnc = fminf(250.E3f, TH.TNO * M_expf(TH_ATO * (T_0 - temp[k])));
                xni = (float)((double)ni[k] + (pni_rfz[k] + pni_wfz[k]) * (double)dtsave);
                pni_inu[k] = 0.5f * (xnc - xni + fabsf(xnc - xni)) * odts;
                pri_inu[k] = fmin((double)rate_max, (double)xm0i * pni_inu[k]);
                pni_inu[k] = pri_inu[k] / (double)xm0i;
            }
            if (L_qi[k]) {
                lami = M_powf(am_i * cig[1] * TH.oig1 * ni[k] / ri[k], TH.obmi);
                ilami = 1. / lami;
                xDi = (float)fmax((double)TH.D0i, (double)(bm_i + mu_i + 1.f) * ilami);
                xmi = am_i * (xDi * xDi * xDi);
                oxmi = 1.f / xmi;
                pri_ide[k] = (double)(C_cube * t1_subl * diffu[k] * ssati[k] * rvs * TH.oig1 * cig[4] * ni[k]) * ilami;
                if (pri_ide[k] < 0.0) {
                    pri_ide[k] = fmax(fmax((double)(-ri[k] * odts), pri_ide[k]), (double)rate_max);
                    pni_ide[k] = pri_ide[k] * (double)oxmi;
                    pni_ide[k] = fmax((double)(-ni[k] * odts), pni_ide[k]);
                } else {
                    pri_ide[k] = fmin(pri_ide[k], (double)rate_max);
                    prs_ide[k] = (1.0 - T2I(tpi_ide)) * pri_ide[k];
                    pri_ide[k] = T2I(tpi_ide) * pri_ide[k];
                }
                if ((idx_i == NTB_I) || (xDi > 5.0f * D0s)) {
                    prs_iau[k] = ri[k] * .99f * odts;
                    pni_iau[k] = ni[k] * .95f * odts;
                } else if (xDi < 0.1f * D0s) {
                    prs_iau[k] = 0.; pni_iau[k] = 0.;
                } else {
                    prs_iau[k] = T2I(tps_iaus) * odts;
                    prs_iau[k] = fmin((double)(ri[k] * .99f * odts), prs_iau[k]);
                    pni_iau[k] = T2I(tni_iaus) * odts;
                    pni_iau[k] = fmin((double)(ni[k] * .95f * odts), pni_iau[k]);
                }
            }
            if (L_qs[k]) {
                C_snow = TH.C_sqrd + (tempc + 15.f) * (TH.C_cubes - TH.C_sqrd) / (-30.f + 15.f);
                C_snow = fmaxf(TH.C_sqrd, fminf(C_snow, TH.C_cubes));
                prs_sde[k] = C_snow * t1_subl * diffu[k] * ssati[k] * rvs
                             * (TH.t1_qs_sd * smo1[k] + TH.t2_qs_sd * rhof2[k] * vsc2[k] * smof[k]);
                if (prs_sde[k] < 0.) prs_sde[k] = fmax(fmax((double)(-rs[k] * odts), prs_sde[k]), (double)rate_max);
                else prs_sde[k] = fmin(prs_sde[k], (double)rate_max);
            }
            if (L_qg[k] && ssati[k] < -eps) {
                prg_gde[k] = (double)(C_cube * t1_subl * diffu[k] * ssati[k] * rvs) * N0_g[k]
                             * ((double)TH.t1_qg_sd * pow(ilamg[k], (double)cge[9])
                                + (double)(TH.t2_qg_sd * vsc2[k] * rhof2[k]) * pow(ilamg[k], (double)cge[10]));
                if (prg_gde[k] < 0.) prg_gde[k] = fmax(fmax((double)(-rg[k] * odts), prg_gde[k]), (double)rate_max);
                else prg_gde[k] = fmin(prg_gde[k], (double)rate_max);
            }
            if (L_qi[k]) {
                lami = M_powf(am_i * cig[1] * TH.oig1 * ni[k] / ri[k], TH.obmi);
                ilami = 1. / lami;
                xDi = (float)fmax((double)TH.D0i, (double)(bm_i + mu_i + 1.f) * ilami);
                xmi = am_i * (xDi * xDi * xDi);
                oxmi = 1.f / xmi;
                if (rs[k] >= TH.r_s[0]) {
                    prs_sci[k] = TH.t1_qs_qi * rhof[k] * TH.Ef_si * ri[k] * smoe[k];
                    pni_sci[k] = prs_sci[k] * (double)oxmi;
                }
                if (rr[k] >= TH.r_r[0] && mvd_r[k] > 4.f * xDi) {
                    lamr = 1. / ilamr[k];
                    pri_rci[k] = (double)(rhof[k] * TH.t1_qr_qi * TH.Ef_ri * ri[k]) * N0_r[k] * pow(lamr + (double)fv_r, -(double)cre[8]);
                    pnr_rci[k] = (double)(rhof[k] * TH.t1_qr_qi * TH.Ef_ri * ni[k]) * N0_r[k] * pow(lamr + (double)fv_r, -(double)cre[8]);
                    pni_rci[k] = pri_rci[k] * (double)oxmi;
                    prr_rci[k] = (double)(rhof[k] * TH.t2_qr_qi * TH.Ef_ri * ni[k]) * N0_r[k] * pow(lamr + (double)fv_r, -(double)cre[7]);
                    prr_rci[k] = fmin((double)(rr[k] * odts), prr_rci[k]);
                    prg_rci[k] = pri_rci[k] + prr_rci[k];
                }
            }
            if (prg_gcw[k] > (double)eps && tempc > -8.0f) {
                tf = 0.f;
                if (tempc >= -5.0f && tempc < -3.0f) tf = 0.5f * (-3.0f - tempc);
                else if (tempc > -8.0f && tempc < -5.0f) tf = 0.33333333f * (8.0f + tempc);
                pni_ihm[k] = (double)(3.5E8f * tf) * prg_gcw[k];
                pri_ihm[k] = (double)xm0i * pni_ihm[k];
                prs_ihm[k] = prs_scw[k] / (prs_scw[k] + prg_gcw[k]) * pri_ihm[k];
                prg_ihm[k] = prg_gcw[k] / (prs_scw[k] + prg_gcw[k]) * pri_ihm[k];
            }
            if (prs_scw[k] > (double)5.0f * prs_sde[k] && prs_sde[k] > (double)eps) {
                r_frac = (float)fmin(30.0, prs_scw[k] / prs_sde[k]);
                g_frac = fminf(0.75f, 0.05f + (r_frac - 5.f) * .028f);
                vts_boost[k] = fminf(1.5f, 1.1f + (r_frac - 5.f) * .016f);
                prg_scw[k] = (double)g_frac * prs_scw[k];
                prs_scw[k] = (double)(1.f - g_frac) * prs_scw[k];
            }
        } else {                  /* :1953-2005 melting */
            if (L_qs[k]) {
                prr_sml[k] = (tempc * tcond[k] - lvap0 * diffu[k] * delQvs[k])
                             * (TH.t1_qs_me * smo1[k] + TH.t2_qs_me * rhof2[k] * vsc2[k] * smof[k]);
                prr_sml[k] = prr_sml[k] + (double)(4218.f * olfus * tempc) * (prr_rcs[k] + prs_scw[k]);
                prr_sml[k] = fmin((double)(rs[k] * odts), fmax(0., prr_sml[k]));
                pnr_sml[k] = (double)(smo0[k] / rs[k]) * prr_sml[k] * (double)M_powf(10.0f, -0.75f * tempc);
                pnr_sml[k] = fmin((double)(smo0[k] * odts), pnr_sml[k]);
                if (tempc > 3.5f || rs[k] < 0.005E-3f) pnr_sml[k] = 0.0;
                if (ssati[k] < 0.f) {
                    prs_sde[k] = TH.C_cubes * t1_subl * diffu[k] * ssati[k] * rvs
                                 * (TH.t1_qs_sd * smo1[k] + TH.t2_qs_sd * rhof2[k] * vsc2[k] * smof[k]);
                    prs_sde[k] = fmax((double)(-rs[k] * odts), prs_sde[k]);
                }
            }
            if (L_qg[k]) {
                prr_gml[k] = (double)(tempc * tcond[k] - lvap0 * diffu[k] * delQvs[k]) * N0_g[k]
                             * ((double)TH.t1_qg_me * pow(ilamg[k], (double)cge[9])
                                + (double)(TH.t2_qg_me * rhof2[k] * vsc2[k]) * pow(ilamg[k], (double)cge[10]));
                prr_gml[k] = fmin((double)(rg[k] * odts), fmax(0., prr_gml[k]));
                pnr_gml[k] = N0_g[k] * (double)cgg[1] * pow(ilamg[k], (double)cge[1]) / (double)rg[k]
                             * prr_gml[k] * (double)M_powf(10.0f, -1.5f * tempc);
                if (tempc > 7.5f || rg[k] < 0.005E-3f) pnr_gml[k] = 0.0;
                if (ssati[k] < 0.f) {
                    prg_gde[k] = (double)(C_cube * t1_subl * diffu[k] * ssati[k] * rvs) * N0_g[k]
                                 * ((double)TH.t1_qg_sd * pow(ilamg[k], (double)cge[9])
                                    + (double)(TH.t2_qg_sd * vsc2[k] * rhof2[k]) * pow(ilamg[k], (double)cge[10]));
                    prg_gde[k] = fmax((double)(-rg[k] * odts), prg_gde[k]);
                }
            }
            if (dt > 120.f) {
                prr_rcw[k] = prr_rcw[k] + prs_scw[k] + prg_gcw[k];
                prs_scw[k] = 0.; prg_gcw[k] = 0.;
            }
        }
    }

#include "thompson_column_part2.inc"
}

/* mp_gt_driver :772-1044.  its..kte are 1-based inclusive like the reference (ims=jms=kms=1).
 * rainncv is read only for SR (never written, as in the reference); snownc/graupelnc/sr may be NULL. */
void orc_thompson(int nx, int nz, int ny, float *qv, float *qc, float *qr, float *qi, float *qs, float *qg, float *ni, float *nr,
                  float *th, const float *pii, const float *p, const float *dz, float dt,
                  float *rainnc, const float *rainncv, float *snownc, float *graupelnc, float *sr,
                  int ids, int ide, int jds, int jde, int kds, int kde, int its, int ite, int jts, int jte, int kts, int kte)
{
    (void)ids; (void)jds; (void)kds; (void)kde;
    const int i_start = its, j_start = jts;
    const int i_end = ite < ide - 1 ? ite : ide - 1;
    const int j_end = jte < jde - 1 ? jte : jde - 1;
    const int nk = kte - kts + 1;
    if (nk > KMAX) { fprintf(stderr, "orc_thompson: nz > KMAX\n"); abort(); }
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = j_start; j <= j_end; ++j)
        for (int i = i_start; i <= i_end; ++i) {
            float qv1d[KMAX], qc1d[KMAX], qi1d[KMAX], qr1d[KMAX], qs1d[KMAX], qg1d[KMAX], ni1d[KMAX], nr1d[KMAX], t1d[KMAX], p1d[KMAX], dz1d[KMAX];
            float pptrain = 0.f, pptsnow = 0.f, pptgraul = 0.f, pptice = 0.f;
            for (int k = kts; k <= kte; ++k) {
                const size_t c = IDX3(i - 1, k - 1, j - 1);
                const int kk = k - kts;
                t1d[kk] = th[c] * pii[c]; p1d[kk] = p[c]; dz1d[kk] = dz[c]; qv1d[kk] = qv[c]; qc1d[kk] = qc[c]; qi1d[kk] = qi[c];
                qr1d[kk] = qr[c]; qs1d[kk] = qs[c]; qg1d[kk] = qg[c]; ni1d[kk] = ni[c]; nr1d[kk] = nr[c];
            }
            th_column(qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, t1d, p1d, dz1d, &pptrain, &pptsnow, &pptgraul, &pptice, nk, dt);
            const size_t c2 = (size_t)(i - 1) + (size_t)nx * (j - 1);
            rainnc[c2] = rainnc[c2] + pptrain + pptsnow + pptgraul + pptice;
            if (snownc) snownc[c2] = snownc[c2] + pptsnow + pptice;
            if (graupelnc) graupelnc[c2] = graupelnc[c2] + pptgraul;
            if (sr) sr[c2] = (pptsnow + pptgraul + pptice) / ((rainncv ? rainncv[c2] : 0.f) + 1.e-12f);
            for (int k = kts; k <= kte; ++k) {
                const size_t c = IDX3(i - 1, k - 1, j - 1);
                const int kk = k - kts;
                qv[c] = qv1d[kk]; qc[c] = qc1d[kk]; qi[c] = qi1d[kk]; qr[c] = qr1d[kk]; qs[c] = qs1d[kk]; qg[c] = qg1d[kk];
                ni[c] = ni1d[kk]; nr[c] = nr1d[kk];
                th[c] = t1d[kk] / pii[c];
                /* :997-1010: the inner re-test reads qv1d again, so the result is always 1e-7 */
                if (qv1d[kk] < 1.E-7f) qv[c] = 1.E-7f;
            }
        }
}
