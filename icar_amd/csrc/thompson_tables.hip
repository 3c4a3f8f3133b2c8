// icar_amd/csrc/thompson_tables.hip -- thompson_init for the device path (row M4).
//
// Reference: src/physics/mp_thompson.f90:342-766 (constants, gamma functions, size bins) and the
// table builders :2853-3578.  The reference spends 56 s on one core building its tables (and caches
// them as Fortran unformatted files); here the O(1e10)-term collection integrals of qr_acr_qg /
// qr_acr_qs run as FP64 HIP kernels (one thread per table entry, the reference's summation order, no
// FMA contraction => bit-identical sums), everything else (a few 1e6 transcendental evaluations) is
// evaluated on the host at init with the host libm exactly like the reference does.
#include "ctx.h"
#include "thompson_state.h"
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
/* ---- numerics :3650-3771 ------------------------------------------------------------------- */
static float th_gammln(float xx)
{   /* :3718-3740 */
    static const double STP = 2.5066282746310005;
    static const double COF[6] = {76.18009172947146, -86.50532032941677, 24.01409824083091,
                                  -1.231739572450155, .1208650973866179e-2, -.5395239384953e-5};
    double x = xx, y = x, tmp = x + 5.5, ser = 1.000000000190015;
    tmp = (x + 0.5) * log(tmp) - tmp;
    for (int j = 0; j < 6; ++j) { y = y + 1.0; ser = ser + COF[j] / y; }
    return (float)(tmp + log(STP * ser / x));
}

static float th_wgamma(float y) { return expf(th_gammln(y)); }   /* :3764-3771 */

static float th_gser(float a, float x)
{   /* :3686-3715 */
    const float gln = th_gammln(a);
    if (x <= 0.f) return 0.f;
    float ap = a, sum = 1.f / a, del = sum;
    for (int n = 1; n <= 100; ++n) {
        ap = ap + 1.f; del = del * x / ap; sum = sum + del;
        if (fabsf(del) < fabsf(sum) * 3.E-7f) break;
    }
    return sum * expf(-x + a * logf(x) - gln);
}

static float th_gcf(float a, float x)
{   /* :3650-3683 */
    const float FPMIN = 1.E-30f;
    const float gln = th_gammln(a);
    float b = x + 1.f - a, c = 1.f / FPMIN, d = 1.f / b, h = d;
    for (int i = 1; i <= 100; ++i) {
        const float an = -i * (i - a);
        b = b + 2.f;
        d = an * d + b; if (fabsf(d) < FPMIN) d = FPMIN;
        c = b + an / c; if (fabsf(c) < FPMIN) c = FPMIN;
        d = 1.f / d;
        const float del = d * c;
        h = h * del;
        if (fabsf(del - 1.f) < 3.E-7f) break;
    }
    return expf(-x + a * logf(x) - gln) * h;
}

static float th_gammp(float a, float x)
{   /* :3743-3761 */
    if (x < 0.f || a <= 0.f) return 0.f;
    if (x < a + 1.f) return th_gser(a, x);
    return 1.f - th_gcf(a, x);
}

/* ---- lookup axes :204-280 ------------------------------------------------------------------- */
static void fill_decades(float *a, int n, float first)
{   /* 1,2,..9 x 10^p sequences exactly as the literals in the source (1.e-6,2.e-6,...) */
    int p = (int)lroundf(log10f(first));
    int idx = 0;
    while (idx < n) {
        for (int m = 1; m <= 9 && idx < n; ++m) {
            char buf[32]; snprintf(buf, sizeof buf, "%d.e%d", m, p);
            a[idx++] = strtof(buf, NULL);
        }
        ++p;
    }
}

static void make_bins(double d0, double dmax, double *D, double *dt)
{   /* :589-634 */
    double xDx[NBINS + 1];
    xDx[0] = d0; xDx[NBINS] = dmax;
    for (int n = 2; n <= NBINS; ++n)
        xDx[n - 1] = exp((double)(n - 1) / (double)NBINS * log(xDx[NBINS] / xDx[0]) + log(xDx[0]));
    for (int n = 0; n < NBINS; ++n) { D[n] = sqrt(xDx[n] * xDx[n + 1]); dt[n] = xDx[n + 1] - xDx[n]; }
}

static inline double pow3(double x) { return x * x * x; }
static inline double pow2(double x) { return x * x; }

static double rain_vt_poly(double Dr)
{   /* :2893-2895 (REAL literals promoted to double) */
    return (double)-0.1021f + (double)4.932E3f * Dr - (double)0.9551E6f * Dr * Dr
         + (double)0.07934E9f * Dr * Dr * Dr - (double)0.002362E12f * Dr * Dr * Dr * Dr;
}

/* ---- table builders ------------------------------------------------------------------------- */
static void table_Efrw(ThState &S)
{   /* :3464-3525 */
    for (int j = 0; j < NBINS; ++j)
        for (int i = 0; i < NBINS; ++i) {
            double Ef_rw = 0.0;
            const double Dr = S.Dr[i], Dc = S.Dc[j];
            const double p = Dc / Dr;
            if (Dr < (double)50.E-6f || Dc < (double)3.E-6f) {
                /* t_Efrw = 0 */
            } else if (p > (double)0.25f) {
                const double X = Dc * 1.e6;
                if (Dr < (double)75.e-6f) Ef_rw = (double)0.026794f * X - (double)0.20604f;
                else if (Dr < (double)125.e-6f) Ef_rw = (double)-0.00066842f * X * X + (double)0.061542f * X - (double)0.37089f;
                else if (Dr < (double)175.e-6f) Ef_rw = (double)4.091e-06f * X * X * X * X - (double)0.00030908f * X * X * X + (double)0.0066237f * X * X - (double)0.0013687f * X - (double)0.073022f;
                else if (Dr < (double)250.e-6f) Ef_rw = (double)9.6719e-5f * X * X * X - (double)0.0068901f * X * X + (double)0.17305f * X - (double)0.65988f;
                else if (Dr < (double)350.e-6f) Ef_rw = (double)9.0488e-5f * X * X * X - (double)0.006585f * X * X + (double)0.16606f * X - (double)0.56125f;
                else Ef_rw = (double)0.00010721f * X * X * X - (double)0.0072962f * X * X + (double)0.1704f * X - (double)0.46929f;
            } else {
                const double vtr = rain_vt_poly(Dr);
                const double stokes = Dc * Dc * vtr * (double)1000.0f / ((double)(9.f * 1.718E-5f) * Dr);
                const double reynolds = (double)9.f * stokes / (p * p * (double)1000.0f);
                const double F = log(reynolds);
                const double G = -0.1007 - 0.358 * F + 0.0261 * F * F;
                const double K0 = exp(G);
                const double z = log(stokes / (K0 + 1.e-15));
                const double H = 0.1465 + 1.302 * z - 0.607 * z * z + 0.293 * z * z * z;
                const double yc0 = 2.0 / (double)TH_PI2 * atan(H);
                Ef_rw = (yc0 + p) * (yc0 + p) / (((double)1.f + p) * ((double)1.f + p));
            }
            float v = fmaxf(0.0f, fminf((float)Ef_rw, 0.95f));
            if (S.Ef_rw_l && Ef_rw != 0.0) v = 1.0f;
            S.t_Efrw[i + NBINS * j] = v;
        }
}

static void table_Efsw(ThState &S)
{   /* :3533-3578 */
    for (int j = 0; j < NBINS; ++j) {
        const double Dc = S.Dc[j];
        const double vtc = 1.19e4 * (1.0e4 * Dc * Dc * 0.25);
        for (int i = 0; i < NBINS; ++i) {
            const double Ds = S.Ds[i];
            const double vts = (double)S.av_s * pow(Ds, (double)S.bv_s) * exp(-(double)S.fv_s * Ds) - vtc;
            const double Ds_m = pow((double)S.am_s * pow(Ds, (double)TH_bm_s) / (double)TH_am_r, (double)S.obmr);
            const double p = Dc / Ds_m;
            float v = 0.0f;
            if (p > (double)0.25f || Ds < (double)TH_D0s || Dc < (double)6.E-6f || vts < (double)1.E-3f) {
                v = 0.0f;
            } else {
                const double stokes = Dc * Dc * vts * (double)1000.0f / ((double)(9.f * 1.718E-5f) * Ds_m);
                const double reynolds = (double)9.f * stokes / (p * p * (double)1000.0f);
                const double F = log(reynolds);
                const double G = -0.1007 - 0.358 * F + 0.0261 * F * F;
                const double K0 = exp(G);
                const double z = log(stokes / (K0 + 1.e-15));
                const double H = 0.1465 + 1.302 * z - 0.607 * z * z + 0.293 * z * z * z;
                const double yc0 = 2.0 / (double)TH_PI2 * atan(H);
                const double Ef_sw = (yc0 + p) * (yc0 + p) / (((double)1.f + p) * ((double)1.f + p));
                v = fmaxf(0.0f, fminf((float)Ef_sw, 0.95f));
                if (S.Ef_sw_l && Ef_sw != 0.0) v = 1.0f;
            }
            S.t_Efsw[i + NBINS * j] = v;
        }
    }
}

static void rain_dist(const ThState &S, int k_n0, int m_r, double *N_r)
{   /* :2921-2926 */
    const double lam_exp = powf(S.N0r_exp[k_n0] * TH_am_r * S.crg[0] / S.r_r[m_r], S.ore1);
    const double lamr = lam_exp * powf(S.crg[2] * S.org2 * S.org1, S.obmr);
    const double N0_r = S.N0r_exp[k_n0] / (S.crg[1] * lam_exp) * pow(lamr, (double)S.cre[1]);
    for (int n2 = 0; n2 < NBINS; ++n2)
        N_r[n2] = N0_r * pow(S.Dr[n2], (double)S.mu_r) * exp(-lamr * S.Dr[n2]) * S.dtr[n2];
}
static double snow_poly(const float *s, float Tc, float b)
{   /* Field et al. (2005) polynomial, evaluated in REAL like :3113-3123 */
    float v = s[0] + s[1] * Tc + s[2] * b + s[3] * Tc * b + s[4] * Tc * Tc + s[5] * b * b + s[6] * Tc * Tc * b
            + s[7] * Tc * b * b + s[8] * Tc * Tc * Tc + s[9] * b * b * b;
    return (double)v;
}

static void freezeH2O(ThState &S)
{   /* :3273-3399 ; tpX_qrfz (ntb_r, ntb_r1, 45), tpi_qcfz (ntb_c, 45) */
    const double orho_w = (double)(1.f / 1000.0f);
    double massr[NBINS], massc[NBINS];
    for (int n2 = 0; n2 < NBINS; ++n2) massr[n2] = (double)TH_am_r * pow3(S.Dr[n2]);
    for (int n = 0; n < NBINS; ++n) massc[n] = (double)TH_am_r * pow3(S.Dc[n]);
    for (int k = 1; k <= 45; ++k) {
        const double Texp = exp((double)k - (double)S.t_adjust * 1.0) - 1.0;
        double N_r[NBINS];
        for (int j = 0; j < NTB_R1; ++j)
            for (int i = 0; i < NTB_R; ++i) {
                const double lam_exp = powf(S.N0r_exp[j] * TH_am_r * S.crg[0] / S.r_r[i], S.ore1);
                const double lamr = lam_exp * powf(S.crg[2] * S.org2 * S.org1, S.obmr);
                const double N0_r = S.N0r_exp[j] / (S.crg[1] * lam_exp) * pow(lamr, (double)S.cre[1]);
                double sum1 = 0, sum2 = 0, sumn1 = 0, sumn2 = 0;
                for (int n2 = NBINS - 1; n2 >= 0; --n2) {
                    N_r[n2] = N0_r * pow(S.Dr[n2], (double)S.mu_r) * exp(-lamr * S.Dr[n2]) * S.dtr[n2];
                    const double vol = massr[n2] * orho_w;
                    double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                    prob = fmax(prob, 0.0);
                    if (massr[n2] < (double)S.xm0g) { sumn1 = sumn1 + prob * N_r[n2]; sum1 = sum1 + prob * N_r[n2] * massr[n2]; }
                    else { sumn2 = sumn2 + prob * N_r[n2]; sum2 = sum2 + prob * N_r[n2] * massr[n2]; }
                    if ((sum1 + sum2) >= (double)S.r_r[i]) break;
                }
                const size_t o = i + NTB_R * (j + NTB_R1 * (size_t)(k - 1));
                S.tpi_qrfz[o] = sum1; S.tni_qrfz[o] = sumn1; S.tpg_qrfz[o] = sum2; S.tnr_qrfz[o] = sumn2;
            }
        for (int i = 0; i < NTB_C; ++i) {
            const double lamc = 1.0e-6 * powf(S.Nt_c * TH_am_r * S.ccg[1] * S.ocg1 / S.r_c[i], S.obmr);
            const double N0_c = 1.0e-18 * S.Nt_c * S.ocg1 * pow(lamc, (double)S.cce[0]);
            double sum1 = 0, sumn2 = 0;
            for (int n = NBINS - 1; n >= 0; --n) {
                const double y = S.Dc[n] * 1.0e6;
                const double vol = massc[n] * orho_w;
                double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                prob = fmax(prob, 0.0);
                double N_c = N0_c * pow(y, (double)S.mu_c) * exp(-lamc * y) * S.dtc[n];
                N_c = 1.0e24 * N_c;
                sumn2 = sumn2 + prob * N_c;
                sum1 = sum1 + prob * N_c * massc[n];
                if (sum1 >= (double)S.r_c[i]) break;
            }
            S.tpi_qcfz[i + NTB_C * (size_t)(k - 1)] = sum1;
            S.tni_qcfz[i + NTB_C * (size_t)(k - 1)] = sumn2;
        }
    }
}

static void qi_aut_qs(ThState &S)
{   /* :3413-3456 ; (ntb_i, ntb_i1) */
    for (int j = 0; j < NTB_I1; ++j)
        for (int i = 0; i < NTB_I; ++i) {
            const double lami = powf(TH_am_i * S.cig[1] * S.oig1 * S.Nt_i[j] / S.r_i[i], S.obmi);
            const double Di_mean = (double)(TH_bm_i + TH_mu_i + 1.f) / lami;
            const double N0_i = (double)(S.Nt_i[j] * S.oig1) * pow(lami, (double)S.cie[0]);
            double t1 = 0, t2 = 0, ide;
            if ((float)Di_mean > 5.f * TH_D0s) { t1 = S.r_i[i]; t2 = S.Nt_i[j]; ide = 0.0; }
            else if ((float)Di_mean < S.D0i) { t1 = 0; t2 = 0; ide = 1.0; }
            else {
                const float xlimit_intg = (float)(lami * (double)TH_D0s);
                ide = (double)th_gammp(TH_mu_i + 2.0f, xlimit_intg) * 1.0;
                for (int n2 = 0; n2 < NBINS; ++n2) {
                    const double N_i = N0_i * pow(S.Di[n2], (double)TH_mu_i) * exp(-lami * S.Di[n2]) * S.dti[n2];
                    if (S.Di[n2] >= (double)TH_D0s) {
                        t1 = t1 + N_i * (double)TH_am_i * pow3(S.Di[n2]);
                        t2 = t2 + N_i;
                    }
                }
            }
            S.tps_iaus[i + NTB_I * j] = t1; S.tni_iaus[i + NTB_I * j] = t2; S.tpi_ide[i + NTB_I * j] = ide;
        }
}

/* ---- thompson_init :342-766 ----------------------------------------------------------------- */
static void th_host_init(ThState &S, const float *p, const int *flags)
{
    S.Nt_c = p[0]; S.TNO = p[1]; S.am_s = p[2]; S.rho_g = p[3]; S.av_s = p[4]; S.bv_s = p[5]; S.fv_s = p[6];
    S.av_g = p[7]; S.bv_g = p[8]; S.av_i = p[9]; S.Ef_si = p[10]; S.Ef_rs = p[11]; S.Ef_rg = p[12]; S.Ef_ri = p[13];
    S.C_cubes = p[14]; S.C_sqrd = p[15]; S.mu_r = p[16]; S.t_adjust = p[17];
    S.Ef_rw_l = flags[0]; S.Ef_sw_l = flags[1];
    S.am_g = TH_PI2 * S.rho_g / 6.0f;
    fill_decades(S.r_c, NTB_C, 1.e-6f); fill_decades(S.r_i, NTB_I, 1.e-10f); fill_decades(S.r_r, NTB_R, 1.e-6f);
    fill_decades(S.r_g, NTB_G, 1.e-5f); fill_decades(S.r_s, NTB_S, 1.e-5f); fill_decades(S.N0r_exp, NTB_R1, 1.e6f);
    fill_decades(S.N0g_exp, NTB_G1, 1.e4f); fill_decades(S.Nt_i, NTB_I1, 1.0f);

    S.mu_c = fminf(15.f, (1000.E6f / S.Nt_c + 2.f));
    S.Sc3 = powf(TH_Sc, 1.f / 3.f);
    S.D0i = powf(TH_xm0i / TH_am_i, 1.f / TH_bm_i);
    S.xm0s = S.am_s * powf(TH_D0s, TH_bm_s);
    S.xm0g = S.am_g * powf(TH_D0g, TH_bm_g);

    float *cce = S.cce, *ccg = S.ccg, *cie = S.cie, *cig = S.cig, *cre = S.cre, *crg = S.crg;
    float *cse = S.cse, *csg = S.csg, *cge = S.cge, *cgg = S.cgg;
    const float mu_c = S.mu_c, mu_r = S.mu_r, bv_s = S.bv_s, bv_g = S.bv_g;
    cce[0] = mu_c + 1.f; cce[1] = TH_bm_r + mu_c + 1.f; cce[2] = TH_bm_r + mu_c + 4.f;
    for (int n = 0; n < 3; ++n) ccg[n] = th_wgamma(cce[n]);
    S.ocg1 = 1.f / ccg[0]; S.ocg2 = 1.f / ccg[1];
    cie[0] = TH_mu_i + 1.f; cie[1] = TH_bm_i + TH_mu_i + 1.f; cie[2] = TH_bm_i + TH_mu_i + TH_bv_i + 1.f;
    cie[3] = TH_mu_i + TH_bv_i + 1.f; cie[4] = TH_mu_i + 2.f; cie[5] = TH_bm_i * 0.5f + TH_mu_i + TH_bv_i + 1.f;
    cie[6] = TH_bm_i * 0.5f + TH_mu_i + 1.f;
    for (int n = 0; n < 7; ++n) cig[n] = th_wgamma(cie[n]);
    S.oig1 = 1.f / cig[0]; S.oig2 = 1.f / cig[1]; S.obmi = 1.f / TH_bm_i;
    cre[0] = TH_bm_r + 1.f; cre[1] = mu_r + 1.f; cre[2] = TH_bm_r + mu_r + 1.f; cre[3] = TH_bm_r * 2.f + mu_r + 1.f;
    cre[4] = mu_r + TH_bv_r + 1.f; cre[5] = TH_bm_r + mu_r + TH_bv_r + 1.f; cre[6] = TH_bm_r * 0.5f + mu_r + TH_bv_r + 1.f;
    cre[7] = TH_bm_r + mu_r + TH_bv_r + 3.f; cre[8] = mu_r + TH_bv_r + 3.f; cre[9] = mu_r + 2.f;
    cre[10] = 0.5f * (TH_bv_r + 5.f + 2.f * mu_r); cre[11] = TH_bm_r * 0.5f + mu_r + 1.f; cre[12] = TH_bm_r * 2.f + mu_r + TH_bv_r + 1.f;
    for (int n = 0; n < 13; ++n) crg[n] = th_wgamma(cre[n]);
    S.obmr = 1.f / TH_bm_r; S.ore1 = 1.f / cre[0]; S.org1 = 1.f / crg[0]; S.org2 = 1.f / crg[1]; S.org3 = 1.f / crg[2];
    cse[0] = TH_bm_s + 1.f; cse[1] = TH_bm_s + 2.f; cse[2] = TH_bm_s * 2.f; cse[3] = TH_bm_s + bv_s + 1.f;
    cse[4] = TH_bm_s * 2.f + bv_s + 1.f; cse[5] = TH_bm_s * 2.f + 1.f; cse[6] = TH_bm_s + TH_mu_s + 1.f;
    cse[7] = TH_bm_s + TH_mu_s + 2.f; cse[8] = TH_bm_s + TH_mu_s + 3.f; cse[9] = TH_bm_s + TH_mu_s + bv_s + 1.f;
    cse[10] = TH_bm_s * 2.f + TH_mu_s + bv_s + 1.f; cse[11] = TH_bm_s * 2.f + TH_mu_s + 1.f; cse[12] = bv_s + 2.f;
    cse[13] = TH_bm_s + bv_s; cse[14] = TH_mu_s + 1.f; cse[15] = 1.0f + (1.0f + bv_s) / 2.f;
    cse[16] = cse[15] + TH_mu_s + 1.f; cse[17] = bv_s + TH_mu_s + 3.f;
    for (int n = 0; n < 18; ++n) csg[n] = th_wgamma(cse[n]);
    S.oams = 1.f / S.am_s; S.obms = 1.f / TH_bm_s; S.ocms = powf(S.oams, S.obms);
    cge[0] = TH_bm_g + 1.f; cge[1] = TH_mu_g + 1.f; cge[2] = TH_bm_g + TH_mu_g + 1.f; cge[3] = TH_bm_g * 2.f + TH_mu_g + 1.f;
    cge[4] = TH_bm_g * 2.f + TH_mu_g + bv_g + 1.f; cge[5] = TH_bm_g + TH_mu_g + bv_g + 1.f; cge[6] = TH_bm_g + TH_mu_g + bv_g + 2.f;
    cge[7] = TH_bm_g + TH_mu_g + bv_g + 3.f; cge[8] = TH_mu_g + bv_g + 3.f; cge[9] = TH_mu_g + 2.f;
    cge[10] = 0.5f * (bv_g + 5.f + 2.f * TH_mu_g); cge[11] = 0.5f * (bv_g + 5.f) + TH_mu_g;
    for (int n = 0; n < 12; ++n) cgg[n] = th_wgamma(cge[n]);
    S.oamg = 1.f / S.am_g; S.obmg = 1.f / TH_bm_g; S.ocmg = powf(S.oamg, S.obmg);
    S.oge1 = 1.f / cge[0]; S.ogg1 = 1.f / cgg[0]; S.ogg2 = 1.f / cgg[1]; S.ogg3 = 1.f / cgg[2];

    /* rate-equation constants :538-568 */
    S.t1_qr_qc = TH_PI2 * .25f * TH_av_r * crg[8];
    S.t1_qr_qi = TH_PI2 * .25f * TH_av_r * crg[8];
    S.t2_qr_qi = TH_PI2 * .25f * TH_am_r * TH_av_r * crg[7];
    S.t1_qg_qc = TH_PI2 * .25f * S.av_g * cgg[8];
    S.t1_qs_qc = TH_PI2 * .25f * S.av_s;
    S.t1_qs_qi = TH_PI2 * .25f * S.av_s;
    S.t1_qr_ev = 0.78f * crg[9];
    S.t2_qr_ev = 0.308f * S.Sc3 * sqrtf(TH_av_r) * crg[10];
    S.t1_qs_sd = 0.86f;
    S.t2_qs_sd = 0.28f * S.Sc3 * sqrtf(S.av_s);
    S.t1_qs_me = TH_PI2 * 4.f * S.C_sqrd * TH_olfus * 0.86f;
    S.t2_qs_me = TH_PI2 * 4.f * S.C_sqrd * TH_olfus * 0.28f * S.Sc3 * sqrtf(S.av_s);
    S.t1_qg_sd = 0.86f * cgg[9];
    S.t2_qg_sd = 0.28f * S.Sc3 * sqrtf(S.av_g) * cgg[10];
    S.t1_qg_me = TH_PI2 * 4.f * TH_C_cube * TH_olfus * 0.86f * cgg[9];
    S.t2_qg_me = TH_PI2 * 4.f * TH_C_cube * TH_olfus * 0.28f * S.Sc3 * sqrtf(S.av_g) * cgg[10];

    /* table index helpers :571-578 */
    S.nic2 = (int)lroundf(log10f(S.r_c[0])); S.nii2 = (int)lroundf(log10f(S.r_i[0])); S.nii3 = (int)lroundf(log10f(S.Nt_i[0]));
    S.nir2 = (int)lroundf(log10f(S.r_r[0])); S.nir3 = (int)lroundf(log10f(S.N0r_exp[0])); S.nis2 = (int)lroundf(log10f(S.r_s[0]));
    S.nig2 = (int)lroundf(log10f(S.r_g[0])); S.nig3 = (int)lroundf(log10f(S.N0g_exp[0]));

    /* size bins :581-634 */
    S.Dc[0] = (double)TH_D0c * 1.0; S.dtc[0] = (double)TH_D0c * 1.0;
    for (int n = 1; n < NBINS; ++n) { S.Dc[n] = S.Dc[n - 1] + 1.0e-6; S.dtc[n] = S.Dc[n] - S.Dc[n - 1]; }
    make_bins((double)S.D0i * 1.0, 5.0 * (double)TH_D0s, S.Di, S.dti);
    make_bins((double)TH_D0r * 1.0, 0.005, S.Dr, S.dtr);
    make_bins((double)TH_D0s * 1.0, 0.02, S.Ds, S.dts);
    make_bins((double)TH_D0g * 1.0, 0.05, S.Dg, S.dtg);

    static const float sa[10] = {5.065339f, -0.062659f, -3.032362f, 0.029469f, -0.000285f, 0.31255f, 0.000204f, 0.003199f, 0.0f, -0.015952f};
    static const float sb[10] = {0.476221f, -0.015896f, 0.165977f, 0.007468f, -0.000141f, 0.060366f, 0.000079f, 0.000594f, 0.0f, -0.003577f};
    static const float Tc[NTB_T] = {-0.01f, -5.f, -10.f, -15.f, -20.f, -25.f, -30.f, -35.f, -40.f};
    memcpy(S.sa, sa, sizeof sa); memcpy(S.sb, sb, sizeof sb); memcpy(S.Tc, Tc, sizeof Tc);
    S.initialized = 1;
}

}  // namespace

// ---- GPU accumulation of the rain x graupel collection integrals (:2937-2972) -------------------
// one thread per (i,j) = (N0g_exp, r_g) entry, one block row per (k,m) = (N0r_exp, r_r) entry.
// Bin-resolved inputs are prepared on the host: N_r[km][n2], N_gT[n][ij] (transposed so that lanes
// read consecutive addresses), fall speeds, masses and diameters.
__global__ void __launch_bounds__(256)
k_qr_acr_qg(const double *__restrict__ N_r, const double *__restrict__ N_gT, const double *__restrict__ Dr,
            const double *__restrict__ Dg, const double *__restrict__ vr, const double *__restrict__ vg,
            const double *__restrict__ massr, const double *__restrict__ massg, float pf, const float *__restrict__ r_r,
            double *tcg_racg, double *tmr_racg, double *tcr_gacr, double *tmg_gacr, double *tnr_racg, double *tnr_gacr)
{
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    const int km = blockIdx.y;
    if (ij >= NTB_G1 * NTB_G) return;
    const int m = km / NTB_R1;
    const double *nr = N_r + (size_t)km * NBINS;
    double t1 = 0, t2 = 0, z1 = 0, z2 = 0, y1 = 0, y2 = 0;
    for (int n2 = 0; n2 < NBINS; ++n2) {
        const double mr = massr[n2], vrr = vr[n2], drr = Dr[n2], nrr = nr[n2];
        for (int n = 0; n < NBINS; ++n) {
            const double ng = N_gT[(size_t)n * (NTB_G1 * NTB_G) + ij];
            const double mg = massg[n];
            const double dvg = 0.5 * ((vrr - vg[n]) + fabs(vrr - vg[n]));
            const double dvr = 0.5 * ((vg[n] - vrr) + fabs(vg[n] - vrr));
            const double s = Dg[n] + drr;
            const double base = (double)pf * s * s;
            t1 = t1 + base * dvg * mg * ng * nrr;
            z1 = z1 + base * dvg * mr * ng * nrr;
            y1 = y1 + base * dvg * ng * nrr;
            t2 = t2 + base * dvr * mr * ng * nrr;
            y2 = y2 + base * dvr * ng * nrr;
            z2 = z2 + base * dvr * mg * ng * nrr;
        }
    }
    const size_t o = (size_t)ij + (size_t)(NTB_G1 * NTB_G) * km;      // (i,j,k,m) Fortran order
    tcg_racg[o] = t1;
    tmr_racg[o] = fmin(z1, (double)r_r[m] * 1.0);
    tcr_gacr[o] = t2; tmg_gacr[o] = z2; tnr_racg[o] = y1; tnr_gacr[o] = y2;
}

// ---- rain x snow (:3152-3217): (i,j) = (r_s, Tc) ------------------------------------------------
__global__ void __launch_bounds__(256)
k_qr_acr_qs(const double *__restrict__ N_r, const double *__restrict__ N_sT, const double *__restrict__ Dr,
            const double *__restrict__ Ds, const double *__restrict__ vr, const double *__restrict__ vs,
            const double *__restrict__ massr, const double *__restrict__ masss, float pf, const float *__restrict__ r_r,
            double *tcs_racs1, double *tmr_racs1, double *tcs_racs2, double *tmr_racs2, double *tcr_sacr1, double *tms_sacr1,
            double *tcr_sacr2, double *tms_sacr2, double *tnr_racs1, double *tnr_racs2, double *tnr_sacr1, double *tnr_sacr2)
{
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    const int km = blockIdx.y;
    if (ij >= NTB_S * NTB_T) return;
    const int m = km / NTB_R1;
    const double *nr = N_r + (size_t)km * NBINS;
    double t1 = 0, t2 = 0, t3 = 0, t4 = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0;
    for (int n2 = 0; n2 < NBINS; ++n2) {
        const double mr = massr[n2], vrr = vr[n2], drr = Dr[n2], nrr = nr[n2];
        for (int n = 0; n < NBINS; ++n) {
            const double ns = N_sT[(size_t)n * (NTB_S * NTB_T) + ij];
            const double ms = masss[n];
            const double dvs = 0.5 * ((vrr - vs[n]) + fabs(vrr - vs[n]));
            const double dvr = 0.5 * ((vs[n] - vrr) + fabs(vs[n] - vrr));
            const double sd = Ds[n] + drr;
            const double base = (double)pf * sd * sd;
            if (mr > (double)1.5f * ms) {
                t1 = t1 + base * dvs * ms * ns * nrr;
                z1 = z1 + base * dvs * mr * ns * nrr;
                y1 = y1 + base * dvs * ns * nrr;
                t2 = t2 + base * dvr * mr * ns * nrr;
                y2 = y2 + base * dvr * ns * nrr;
                z2 = z2 + base * dvr * ms * ns * nrr;
            } else {
                t3 = t3 + base * dvs * ms * ns * nrr;
                z3 = z3 + base * dvs * mr * ns * nrr;
                y3 = y3 + base * dvs * ns * nrr;
                t4 = t4 + base * dvr * mr * ns * nrr;
                y4 = y4 + base * dvr * ns * nrr;
                z4 = z4 + base * dvr * ms * ns * nrr;
            }
        }
    }
    const size_t o = (size_t)ij + (size_t)(NTB_S * NTB_T) * km;
    tcs_racs1[o] = t1; tmr_racs1[o] = fmin(z1, (double)r_r[m] * 1.0);
    tcs_racs2[o] = t3; tmr_racs2[o] = z3;
    tcr_sacr1[o] = t2; tms_sacr1[o] = z2;
    tcr_sacr2[o] = t4; tms_sacr2[o] = z4;
    tnr_racs1[o] = y1; tnr_racs2[o] = y3;
    tnr_sacr1[o] = y2; tnr_sacr2[o] = y4;
}

// ---- host-side bin-resolved distributions feeding the two kernels --------------------------------
namespace {
void graupel_dist(const ThState &S, int i, int j, double *N_g)
{   // :2930-2935
    const double lam_exp = powf(S.N0g_exp[i] * S.am_g * S.cgg[0] / S.r_g[j], S.oge1);
    const double lamg = lam_exp * powf(S.cgg[2] * S.ogg2 * S.ogg1, S.obmg);
    const double N0_g = S.N0g_exp[i] / (S.cgg[1] * lam_exp) * pow(lamg, (double)S.cge[1]);
    for (int n = 0; n < NBINS; ++n)
        N_g[n] = N0_g * pow(S.Dg[n], (double)TH_mu_g) * exp(-lamg * S.Dg[n]) * S.dtg[n];
}

void snow_dist(const ThState &S, int i, int j, double *N_s)
{   // :3111-3150
    const double M2 = (double)(S.r_s[i] * S.oams) * 1.0;
    double second;
    if (TH_bm_s > 2.0f - 1.E-3f && TH_bm_s < 2.0f + 1.E-3f) {
        const double loga_ = snow_poly(S.sa, S.Tc[j], TH_bm_s);
        const double a_ = pow(10.0, loga_);
        const double b_ = snow_poly(S.sb, S.Tc[j], TH_bm_s);
        second = pow(M2 / a_, 1. / b_);
    } else second = M2;
    const double loga_ = snow_poly(S.sa, S.Tc[j], S.cse[0]);
    const double a_ = pow(10.0, loga_);
    const double b_ = snow_poly(S.sb, S.Tc[j], S.cse[0]);
    const double M3 = a_ * pow(second, b_);
    const double oM3 = 1. / M3;
    const double Mrat = M2 * (M2 * oM3) * (M2 * oM3) * (M2 * oM3);
    const double M0 = pow(M2 * oM3, (double)TH_mu_s);
    const double slam1 = M2 * oM3 * (double)TH_Lam0;
    const double slam2 = M2 * oM3 * (double)TH_Lam1;
    for (int n = 0; n < NBINS; ++n)
        N_s[n] = Mrat * ((double)TH_Kap0 * exp(-slam1 * S.Ds[n])
                 + (double)TH_Kap1 * M0 * pow(S.Ds[n], (double)TH_mu_s) * exp(-slam2 * S.Ds[n])) * S.dts[n];
}

template <typename T> T *to_device(icar_hip_ctx *c, std::vector<void *> &allocs, const T *h, size_t n)
{
    T *d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return nullptr;
    allocs.push_back(d);
    if (h) hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, c->stream);
    else hipMemsetAsync(d, 0, n * sizeof(T), c->stream);
    return d;
}
}  // namespace

struct ThompsonTables {
    ThState host;               // constants + host copies of the small tables
    ThState *d_state = nullptr; // device struct (same layout, device table pointers)
    ThState dev_view;           // host copy of the device struct (for table downloads)
    std::vector<void *> allocs;
    std::vector<double *> host_tabs;
};

void icar_thompson_free(icar_hip_ctx *c)
{
    if (!c->thompson) return;
    for (void *p : c->thompson->allocs) hipFree(p);
    for (double *p : c->thompson->host_tabs) free(p);
    delete c->thompson;
    c->thompson = nullptr;
}

const ThState *icar_thompson_device_state(icar_hip_ctx *c) { return c->thompson ? c->thompson->d_state : nullptr; }
const ThState *icar_thompson_host_state(icar_hip_ctx *c) { return c->thompson ? &c->thompson->host : nullptr; }

int icar_thompson_init_run(icar_hip_ctx *c, const float *params, const int *flags)
{
    icar_thompson_free(c);
    ThompsonTables *T = new ThompsonTables();
    c->thompson = T;
    ThState &S = T->host;
    memset(&S, 0, sizeof S);
    th_host_init(S, params, flags);

    // ---- small tables on the host (:3273-3578) ----
    auto halloc = [&](size_t n) { double *p = (double *)calloc(n, sizeof(double)); T->host_tabs.push_back(p); return p; };
    S.tpi_qcfz = halloc(NTB_C * 45); S.tni_qcfz = halloc(NTB_C * 45);
    S.tpi_qrfz = halloc(NTB_R * NTB_R1 * 45); S.tpg_qrfz = halloc(NTB_R * NTB_R1 * 45);
    S.tni_qrfz = halloc(NTB_R * NTB_R1 * 45); S.tnr_qrfz = halloc(NTB_R * NTB_R1 * 45);
    S.tps_iaus = halloc(NTB_I * NTB_I1); S.tni_iaus = halloc(NTB_I * NTB_I1); S.tpi_ide = halloc(NTB_I * NTB_I1);
    S.t_Efrw = halloc(NBINS * NBINS); S.t_Efsw = halloc(NBINS * NBINS);
    table_Efrw(S); table_Efsw(S); freezeH2O(S); qi_aut_qs(S);

    // ---- bin-resolved inputs of the two collection integrals ----
    const int NKM = NTB_R * NTB_R1, NIJG = NTB_G1 * NTB_G, NIJS = NTB_S * NTB_T;
    std::vector<double> N_r((size_t)NKM * NBINS), N_gT((size_t)NBINS * NIJG), N_sT((size_t)NBINS * NIJS);
    std::vector<double> vr(NBINS), vg(NBINS), vs(NBINS), massr(NBINS), massg(NBINS), masss(NBINS), tmp(NBINS);
    for (int n = 0; n < NBINS; ++n) {
        vr[n] = rain_vt_poly(S.Dr[n]);
        vg[n] = (double)S.av_g * pow(S.Dg[n], (double)S.bv_g);
        vs[n] = (double)(1.5f * S.av_s) * pow(S.Ds[n], (double)S.bv_s) * exp(-(double)S.fv_s * S.Ds[n]);
        massr[n] = (double)TH_am_r * pow3(S.Dr[n]);
        massg[n] = (double)S.am_g * pow3(S.Dg[n]);
        masss[n] = (double)S.am_s * pow2(S.Ds[n]);
    }
    for (int km = 0; km < NKM; ++km) rain_dist(S, km % NTB_R1, km / NTB_R1, &N_r[(size_t)km * NBINS]);
    for (int j = 0; j < NTB_G; ++j)
        for (int i = 0; i < NTB_G1; ++i) {
            graupel_dist(S, i, j, tmp.data());
            for (int n = 0; n < NBINS; ++n) N_gT[(size_t)n * NIJG + (i + NTB_G1 * j)] = tmp[n];
        }
    for (int j = 0; j < NTB_T; ++j)
        for (int i = 0; i < NTB_S; ++i) {
            snow_dist(S, i, j, tmp.data());
            for (int n = 0; n < NBINS; ++n) N_sT[(size_t)n * NIJS + (i + NTB_S * j)] = tmp[n];
        }

    // ---- device: inputs, tables, kernels ----
    auto &A = T->allocs;
    double *dNr = to_device(c, A, N_r.data(), N_r.size()), *dNg = to_device(c, A, N_gT.data(), N_gT.size());
    double *dNs = to_device(c, A, N_sT.data(), N_sT.size());
    double *dDr = to_device(c, A, S.Dr, NBINS), *dDg = to_device(c, A, S.Dg, NBINS), *dDs = to_device(c, A, S.Ds, NBINS);
    double *dvr = to_device(c, A, vr.data(), NBINS), *dvg = to_device(c, A, vg.data(), NBINS), *dvs = to_device(c, A, vs.data(), NBINS);
    double *dmr = to_device(c, A, massr.data(), NBINS), *dmg = to_device(c, A, massg.data(), NBINS), *dms = to_device(c, A, masss.data(), NBINS);
    float *drr = to_device(c, A, S.r_r, NTB_R);
    ThState D = S;
    const size_t n4g = (size_t)NIJG * NKM, n4s = (size_t)NIJS * NKM;
    double **g6[] = {&D.tcg_racg, &D.tmr_racg, &D.tcr_gacr, &D.tmg_gacr, &D.tnr_racg, &D.tnr_gacr};
    for (auto p : g6) *p = to_device<double>(c, A, nullptr, n4g);
    double **s12[] = {&D.tcs_racs1, &D.tmr_racs1, &D.tcs_racs2, &D.tmr_racs2, &D.tcr_sacr1, &D.tms_sacr1,
                      &D.tcr_sacr2, &D.tms_sacr2, &D.tnr_racs1, &D.tnr_racs2, &D.tnr_sacr1, &D.tnr_sacr2};
    for (auto p : s12) *p = to_device<double>(c, A, nullptr, n4s);
    D.tpi_qcfz = to_device(c, A, S.tpi_qcfz, NTB_C * 45); D.tni_qcfz = to_device(c, A, S.tni_qcfz, NTB_C * 45);
    D.tpi_qrfz = to_device(c, A, S.tpi_qrfz, NTB_R * NTB_R1 * 45); D.tpg_qrfz = to_device(c, A, S.tpg_qrfz, NTB_R * NTB_R1 * 45);
    D.tni_qrfz = to_device(c, A, S.tni_qrfz, NTB_R * NTB_R1 * 45); D.tnr_qrfz = to_device(c, A, S.tnr_qrfz, NTB_R * NTB_R1 * 45);
    D.tps_iaus = to_device(c, A, S.tps_iaus, NTB_I * NTB_I1); D.tni_iaus = to_device(c, A, S.tni_iaus, NTB_I * NTB_I1);
    D.tpi_ide = to_device(c, A, S.tpi_ide, NTB_I * NTB_I1);
    D.t_Efrw = to_device(c, A, S.t_Efrw, NBINS * NBINS); D.t_Efsw = to_device(c, A, S.t_Efsw, NBINS * NBINS);
    for (void *p : A) if (!p) { icar_set_error("thompson_init: hipMalloc failed"); return 1; }
    if (!dNr || !dNg || !dNs || !drr || !D.t_Efsw) { icar_set_error("thompson_init: hipMalloc failed"); return 1; }

    {
        dim3 b(256), g((NIJG + 255) / 256, NKM);
        hipLaunchKernelGGL(k_qr_acr_qg, g, b, 0, c->stream, dNr, dNg, dDr, dDg, dvr, dvg, dmr, dmg, TH_PI2 * .25f * S.Ef_rg, drr,
                           D.tcg_racg, D.tmr_racg, D.tcr_gacr, D.tmg_gacr, D.tnr_racg, D.tnr_gacr);
        dim3 g2((NIJS + 255) / 256, NKM);
        hipLaunchKernelGGL(k_qr_acr_qs, g2, b, 0, c->stream, dNr, dNs, dDr, dDs, dvr, dvs, dmr, dms, TH_PI2 * .25f * S.Ef_rs, drr,
                           D.tcs_racs1, D.tmr_racs1, D.tcs_racs2, D.tmr_racs2, D.tcr_sacr1, D.tms_sacr1,
                           D.tcr_sacr2, D.tms_sacr2, D.tnr_racs1, D.tnr_racs2, D.tnr_sacr1, D.tnr_sacr2);
        HIPCHK(hipGetLastError());
    }
    T->dev_view = D;
    T->d_state = to_device(c, A, &D, 1);
    if (!T->d_state) { icar_set_error("thompson_init: hipMalloc failed"); return 1; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return icar_thompson_prepare_constants(c);
}

// Download one lookup table by its reference name (tests / cross-checks with ICAR's own *.dat caches).
int icar_thompson_table_download(icar_hip_ctx *c, const char *name, double *out, size_t cap, size_t *n_out)
{
    if (!c->thompson) { icar_set_error("thompson tables are not initialised"); return 1; }
    const ThState &D = c->thompson->dev_view;
    const size_t n4g = (size_t)NTB_G1 * NTB_G * NTB_R1 * NTB_R, n4s = (size_t)NTB_S * NTB_T * NTB_R1 * NTB_R;
    const size_t n3 = (size_t)NTB_R * NTB_R1 * 45, n2c = (size_t)NTB_C * 45, n2i = (size_t)NTB_I * NTB_I1, n2e = (size_t)NBINS * NBINS;
    const double *src = nullptr; size_t n = 0;
#define TT(nm, cnt) if (!strcmp(name, #nm)) { src = D.nm; n = cnt; }
    TT(tcg_racg, n4g) TT(tmr_racg, n4g) TT(tcr_gacr, n4g) TT(tmg_gacr, n4g) TT(tnr_racg, n4g) TT(tnr_gacr, n4g)
    TT(tcs_racs1, n4s) TT(tmr_racs1, n4s) TT(tcs_racs2, n4s) TT(tmr_racs2, n4s) TT(tcr_sacr1, n4s) TT(tms_sacr1, n4s)
    TT(tcr_sacr2, n4s) TT(tms_sacr2, n4s) TT(tnr_racs1, n4s) TT(tnr_racs2, n4s) TT(tnr_sacr1, n4s) TT(tnr_sacr2, n4s)
    TT(tpi_qcfz, n2c) TT(tni_qcfz, n2c) TT(tpi_qrfz, n3) TT(tpg_qrfz, n3) TT(tni_qrfz, n3) TT(tnr_qrfz, n3)
    TT(tps_iaus, n2i) TT(tni_iaus, n2i) TT(tpi_ide, n2i) TT(t_Efrw, n2e) TT(t_Efsw, n2e)
#undef TT
    if (!src) { icar_set_error(std::string("unknown thompson table ") + name); return 1; }
    if (n_out) *n_out = n;
    if (!out) return 0;
    if (cap < n) { icar_set_error("thompson table: output buffer too small"); return 1; }
    HIPCHK(hipMemcpyAsync(out, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
