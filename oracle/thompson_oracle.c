/* oracle/thompson_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C CPU restatement of the Thompson et al. (2008) bulk microphysics as shipped in the
 * reference: src/physics/mp_thompson.f90 (thompson_init :342-766, mp_gt_driver :772-1044,
 * mp_thompson :1057-2844, table builders :2853-3578, GAMMLN/GAMMP/WGAMMA :3650-3771,
 * RSLF/RSIF :3776-3835).  REAL -> float, DOUBLE PRECISION -> double, following the reference's
 * mixed-precision expressions.  Pinned against the compiled reference (oracle/_ref) by
 * tests/test_oracle_vs_ref.py and the committed fixtures in tests/golden/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "thompson_oracle.h"

struct thompson_state TH;

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

static const float th_sa[10] = {5.065339f, -0.062659f, -3.032362f, 0.029469f, -0.000285f, 0.31255f, 0.000204f, 0.003199f, 0.0f, -0.015952f};
static const float th_sb[10] = {0.476221f, -0.015896f, 0.165977f, 0.007468f, -0.000141f, 0.060366f, 0.000079f, 0.000594f, 0.0f, -0.003577f};
static const float th_Tc[NTB_T] = {-0.01f, -5.f, -10.f, -15.f, -20.f, -25.f, -30.f, -35.f, -40.f};

const float *th_sa_ptr(void) { return th_sa; }
const float *th_sb_ptr(void) { return th_sb; }

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
static void table_Efrw(void)
{   /* :3464-3525 */
    for (int j = 0; j < NBINS; ++j)
        for (int i = 0; i < NBINS; ++i) {
            double Ef_rw = 0.0;
            const double Dr = TH.Dr[i], Dc = TH.Dc[j];
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
            if (TH.Ef_rw_l && Ef_rw != 0.0) v = 1.0f;
            TH.t_Efrw[i + NBINS * j] = v;
        }
}

static void table_Efsw(void)
{   /* :3533-3578 */
    for (int j = 0; j < NBINS; ++j) {
        const double Dc = TH.Dc[j];
        const double vtc = 1.19e4 * (1.0e4 * Dc * Dc * 0.25);
        for (int i = 0; i < NBINS; ++i) {
            const double Ds = TH.Ds[i];
            const double vts = (double)TH.av_s * pow(Ds, (double)TH.bv_s) * exp(-(double)TH.fv_s * Ds) - vtc;
            const double Ds_m = pow((double)TH.am_s * pow(Ds, (double)TH_bm_s) / (double)TH_am_r, (double)TH.obmr);
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
                if (TH.Ef_sw_l && Ef_sw != 0.0) v = 1.0f;
            }
            TH.t_Efsw[i + NBINS * j] = v;
        }
    }
}

static void rain_dist(int k_n0, int m_r, double *N_r)
{   /* :2921-2926 */
    const double lam_exp = powf(TH.N0r_exp[k_n0] * TH_am_r * TH.crg[0] / TH.r_r[m_r], TH.ore1);
    const double lamr = lam_exp * powf(TH.crg[2] * TH.org2 * TH.org1, TH.obmr);
    const double N0_r = TH.N0r_exp[k_n0] / (TH.crg[1] * lam_exp) * pow(lamr, (double)TH.cre[1]);
    for (int n2 = 0; n2 < NBINS; ++n2)
        N_r[n2] = N0_r * pow(TH.Dr[n2], (double)TH.mu_r) * exp(-lamr * TH.Dr[n2]) * TH.dtr[n2];
}

static void qr_acr_qg(void)
{   /* :2853-3007 ; tables (ntb_g1, ntb_g, ntb_r1, ntb_r) Fortran order */
    double vr[NBINS], vg[NBINS];
    for (int n2 = 0; n2 < NBINS; ++n2) vr[n2] = rain_vt_poly(TH.Dr[n2]);
    for (int n = 0; n < NBINS; ++n) vg[n] = (double)TH.av_g * pow(TH.Dg[n], (double)TH.bv_g);
    const float pf = TH_PI2 * .25f * TH.Ef_rg;
#pragma omp parallel for schedule(dynamic, 1)
    for (int km = 0; km < NTB_R * NTB_R1; ++km) {
        const int m = km / NTB_R1, k = km % NTB_R1;
        double N_r[NBINS], N_g[NBINS];
        rain_dist(k, m, N_r);
        for (int j = 0; j < NTB_G; ++j)
            for (int i = 0; i < NTB_G1; ++i) {
                const double lam_exp = powf(TH.N0g_exp[i] * TH.am_g * TH.cgg[0] / TH.r_g[j], TH.oge1);
                const double lamg = lam_exp * powf(TH.cgg[2] * TH.ogg2 * TH.ogg1, TH.obmg);
                const double N0_g = TH.N0g_exp[i] / (TH.cgg[1] * lam_exp) * pow(lamg, (double)TH.cge[1]);
                for (int n = 0; n < NBINS; ++n)
                    N_g[n] = N0_g * pow(TH.Dg[n], (double)TH_mu_g) * exp(-lamg * TH.Dg[n]) * TH.dtg[n];
                double t1 = 0, t2 = 0, z1 = 0, z2 = 0, y1 = 0, y2 = 0;
                for (int n2 = 0; n2 < NBINS; ++n2) {
                    const double massr = (double)TH_am_r * pow3(TH.Dr[n2]);
                    for (int n = 0; n < NBINS; ++n) {
                        const double massg = (double)TH.am_g * pow3(TH.Dg[n]);
                        const double dvg = 0.5 * ((vr[n2] - vg[n]) + fabs(vr[n2] - vg[n]));
                        const double dvr = 0.5 * ((vg[n] - vr[n2]) + fabs(vg[n] - vr[n2]));
                        const double s = TH.Dg[n] + TH.Dr[n2];
                        const double base = (double)pf * s * s;
                        t1 = t1 + base * dvg * massg * N_g[n] * N_r[n2];
                        z1 = z1 + base * dvg * massr * N_g[n] * N_r[n2];
                        y1 = y1 + base * dvg * N_g[n] * N_r[n2];
                        t2 = t2 + base * dvr * massr * N_g[n] * N_r[n2];
                        y2 = y2 + base * dvr * N_g[n] * N_r[n2];
                        z2 = z2 + base * dvr * massg * N_g[n] * N_r[n2];
                    }
                }
                const size_t o = i + NTB_G1 * (j + NTB_G * ((size_t)k + NTB_R1 * m));
                TH.tcg_racg[o] = t1;
                TH.tmr_racg[o] = fmin(z1, (double)TH.r_r[m] * 1.0);
                TH.tcr_gacr[o] = t2;
                TH.tmg_gacr[o] = z2;
                TH.tnr_racg[o] = y1;
                TH.tnr_gacr[o] = y2;
            }
    }
}

static double snow_poly(const float *s, float Tc, float b)
{   /* Field et al. (2005) polynomial, evaluated in REAL like :3113-3123 */
    float v = s[0] + s[1] * Tc + s[2] * b + s[3] * Tc * b + s[4] * Tc * Tc + s[5] * b * b + s[6] * Tc * Tc * b
            + s[7] * Tc * b * b + s[8] * Tc * Tc * Tc + s[9] * b * b * b;
    return (double)v;
}

static void qr_acr_qs(void)
{   /* :3014-3264 ; tables (ntb_s, ntb_t, ntb_r1, ntb_r) */
    double vr[NBINS], vs[NBINS];
    for (int n2 = 0; n2 < NBINS; ++n2) vr[n2] = rain_vt_poly(TH.Dr[n2]);
    for (int n = 0; n < NBINS; ++n) vs[n] = (double)(1.5f * TH.av_s) * pow(TH.Ds[n], (double)TH.bv_s) * exp(-(double)TH.fv_s * TH.Ds[n]);
    const float pf = TH_PI2 * .25f * TH.Ef_rs;
#pragma omp parallel for schedule(dynamic, 1)
    for (int km = 0; km < NTB_R * NTB_R1; ++km) {
        const int m = km / NTB_R1, k = km % NTB_R1;
        double N_r[NBINS], N_s[NBINS];
        rain_dist(k, m, N_r);
        for (int j = 0; j < NTB_T; ++j)
            for (int i = 0; i < NTB_S; ++i) {
                const double M2 = (double)(TH.r_s[i] * TH.oams) * 1.0;
                double second;
                if (TH_bm_s > 2.0f - 1.E-3f && TH_bm_s < 2.0f + 1.E-3f) {
                    const double loga_ = snow_poly(th_sa, th_Tc[j], TH_bm_s);
                    const double a_ = pow(10.0, loga_);
                    const double b_ = snow_poly(th_sb, th_Tc[j], TH_bm_s);
                    second = pow(M2 / a_, 1. / b_);
                } else second = M2;
                const double loga_ = snow_poly(th_sa, th_Tc[j], TH.cse[0]);
                const double a_ = pow(10.0, loga_);
                const double b_ = snow_poly(th_sb, th_Tc[j], TH.cse[0]);
                const double M3 = a_ * pow(second, b_);
                const double oM3 = 1. / M3;
                const double Mrat = M2 * (M2 * oM3) * (M2 * oM3) * (M2 * oM3);
                const double M0 = pow(M2 * oM3, (double)TH_mu_s);
                const double slam1 = M2 * oM3 * (double)TH_Lam0;
                const double slam2 = M2 * oM3 * (double)TH_Lam1;
                for (int n = 0; n < NBINS; ++n)
                    N_s[n] = Mrat * ((double)TH_Kap0 * exp(-slam1 * TH.Ds[n])
                             + (double)TH_Kap1 * M0 * pow(TH.Ds[n], (double)TH_mu_s) * exp(-slam2 * TH.Ds[n])) * TH.dts[n];
                double t1 = 0, t2 = 0, t3 = 0, t4 = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0;
                for (int n2 = 0; n2 < NBINS; ++n2) {
                    const double massr = (double)TH_am_r * pow3(TH.Dr[n2]);
                    for (int n = 0; n < NBINS; ++n) {
                        const double masss = (double)TH.am_s * pow(TH.Ds[n], (double)TH_bm_s);
                        const double dvs = 0.5 * ((vr[n2] - vs[n]) + fabs(vr[n2] - vs[n]));
                        const double dvr = 0.5 * ((vs[n] - vr[n2]) + fabs(vs[n] - vr[n2]));
                        const double sd = TH.Ds[n] + TH.Dr[n2];
                        const double base = (double)pf * sd * sd;
                        if (massr > (double)1.5f * masss) {
                            t1 = t1 + base * dvs * masss * N_s[n] * N_r[n2];
                            z1 = z1 + base * dvs * massr * N_s[n] * N_r[n2];
                            y1 = y1 + base * dvs * N_s[n] * N_r[n2];
                            t2 = t2 + base * dvr * massr * N_s[n] * N_r[n2];
                            y2 = y2 + base * dvr * N_s[n] * N_r[n2];
                            z2 = z2 + base * dvr * masss * N_s[n] * N_r[n2];
                        } else {
                            t3 = t3 + base * dvs * masss * N_s[n] * N_r[n2];
                            z3 = z3 + base * dvs * massr * N_s[n] * N_r[n2];
                            y3 = y3 + base * dvs * N_s[n] * N_r[n2];
                            t4 = t4 + base * dvr * massr * N_s[n] * N_r[n2];
                            y4 = y4 + base * dvr * N_s[n] * N_r[n2];
                            z4 = z4 + base * dvr * masss * N_s[n] * N_r[n2];
                        }
                    }
                }
                const size_t o = i + NTB_S * (j + NTB_T * ((size_t)k + NTB_R1 * m));
                TH.tcs_racs1[o] = t1; TH.tmr_racs1[o] = fmin(z1, (double)TH.r_r[m] * 1.0);
                TH.tcs_racs2[o] = t3; TH.tmr_racs2[o] = z3;
                TH.tcr_sacr1[o] = t2; TH.tms_sacr1[o] = z2;
                TH.tcr_sacr2[o] = t4; TH.tms_sacr2[o] = z4;
                TH.tnr_racs1[o] = y1; TH.tnr_racs2[o] = y3;
                TH.tnr_sacr1[o] = y2; TH.tnr_sacr2[o] = y4;
            }
    }
}

static void freezeH2O(void)
{   /* :3273-3399 ; tpX_qrfz (ntb_r, ntb_r1, 45), tpi_qcfz (ntb_c, 45) */
    const double orho_w = (double)(1.f / 1000.0f);
    double massr[NBINS], massc[NBINS];
    for (int n2 = 0; n2 < NBINS; ++n2) massr[n2] = (double)TH_am_r * pow3(TH.Dr[n2]);
    for (int n = 0; n < NBINS; ++n) massc[n] = (double)TH_am_r * pow3(TH.Dc[n]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 1; k <= 45; ++k) {
        const double Texp = exp((double)k - (double)TH.t_adjust * 1.0) - 1.0;
        double N_r[NBINS];
        for (int j = 0; j < NTB_R1; ++j)
            for (int i = 0; i < NTB_R; ++i) {
                const double lam_exp = powf(TH.N0r_exp[j] * TH_am_r * TH.crg[0] / TH.r_r[i], TH.ore1);
                const double lamr = lam_exp * powf(TH.crg[2] * TH.org2 * TH.org1, TH.obmr);
                const double N0_r = TH.N0r_exp[j] / (TH.crg[1] * lam_exp) * pow(lamr, (double)TH.cre[1]);
                double sum1 = 0, sum2 = 0, sumn1 = 0, sumn2 = 0;
                for (int n2 = NBINS - 1; n2 >= 0; --n2) {
                    N_r[n2] = N0_r * pow(TH.Dr[n2], (double)TH.mu_r) * exp(-lamr * TH.Dr[n2]) * TH.dtr[n2];
                    const double vol = massr[n2] * orho_w;
                    double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                    prob = fmax(prob, 0.0);
                    if (massr[n2] < (double)TH.xm0g) { sumn1 = sumn1 + prob * N_r[n2]; sum1 = sum1 + prob * N_r[n2] * massr[n2]; }
                    else { sumn2 = sumn2 + prob * N_r[n2]; sum2 = sum2 + prob * N_r[n2] * massr[n2]; }
                    if ((sum1 + sum2) >= (double)TH.r_r[i]) break;
                }
                const size_t o = i + NTB_R * (j + NTB_R1 * (size_t)(k - 1));
                TH.tpi_qrfz[o] = sum1; TH.tni_qrfz[o] = sumn1; TH.tpg_qrfz[o] = sum2; TH.tnr_qrfz[o] = sumn2;
            }
        for (int i = 0; i < NTB_C; ++i) {
            const double lamc = 1.0e-6 * powf(TH.Nt_c * TH_am_r * TH.ccg[1] * TH.ocg1 / TH.r_c[i], TH.obmr);
            const double N0_c = 1.0e-18 * TH.Nt_c * TH.ocg1 * pow(lamc, (double)TH.cce[0]);
            double sum1 = 0, sumn2 = 0;
            for (int n = NBINS - 1; n >= 0; --n) {
                const double y = TH.Dc[n] * 1.0e6;
                const double vol = massc[n] * orho_w;
                double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                prob = fmax(prob, 0.0);
                double N_c = N0_c * pow(y, (double)TH.mu_c) * exp(-lamc * y) * TH.dtc[n];
                N_c = 1.0e24 * N_c;
                sumn2 = sumn2 + prob * N_c;
                sum1 = sum1 + prob * N_c * massc[n];
                if (sum1 >= (double)TH.r_c[i]) break;
            }
            TH.tpi_qcfz[i + NTB_C * (size_t)(k - 1)] = sum1;
            TH.tni_qcfz[i + NTB_C * (size_t)(k - 1)] = sumn2;
        }
    }
}

static void qi_aut_qs(void)
{   /* :3413-3456 ; (ntb_i, ntb_i1) */
    for (int j = 0; j < NTB_I1; ++j)
        for (int i = 0; i < NTB_I; ++i) {
            const double lami = powf(TH_am_i * TH.cig[1] * TH.oig1 * TH.Nt_i[j] / TH.r_i[i], TH.obmi);
            const double Di_mean = (double)(TH_bm_i + TH_mu_i + 1.f) / lami;
            const double N0_i = (double)(TH.Nt_i[j] * TH.oig1) * pow(lami, (double)TH.cie[0]);
            double t1 = 0, t2 = 0, ide;
            if ((float)Di_mean > 5.f * TH_D0s) { t1 = TH.r_i[i]; t2 = TH.Nt_i[j]; ide = 0.0; }
            else if ((float)Di_mean < TH.D0i) { t1 = 0; t2 = 0; ide = 1.0; }
            else {
                const float xlimit_intg = (float)(lami * (double)TH_D0s);
                ide = (double)th_gammp(TH_mu_i + 2.0f, xlimit_intg) * 1.0;
                for (int n2 = 0; n2 < NBINS; ++n2) {
                    const double N_i = N0_i * pow(TH.Di[n2], (double)TH_mu_i) * exp(-lami * TH.Di[n2]) * TH.dti[n2];
                    if (TH.Di[n2] >= (double)TH_D0s) {
                        t1 = t1 + N_i * (double)TH_am_i * pow3(TH.Di[n2]);
                        t2 = t2 + N_i;
                    }
                }
            }
            TH.tps_iaus[i + NTB_I * j] = t1; TH.tni_iaus[i + NTB_I * j] = t2; TH.tpi_ide[i + NTB_I * j] = ide;
        }
}

/* ---- thompson_init :342-766 ----------------------------------------------------------------- */
static double *dalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }

void orc_thompson_init(const float *p, const int *flags, int build_tables)
{
    if (!TH.tcg_racg) {
        const size_t n4g = (size_t)NTB_G1 * NTB_G * NTB_R1 * NTB_R, n4s = (size_t)NTB_S * NTB_T * NTB_R1 * NTB_R;
        double **g[] = {&TH.tcg_racg, &TH.tmr_racg, &TH.tcr_gacr, &TH.tmg_gacr, &TH.tnr_racg, &TH.tnr_gacr};
        for (int i = 0; i < 6; ++i) *g[i] = dalloc(n4g);
        double **s[] = {&TH.tcs_racs1, &TH.tmr_racs1, &TH.tcs_racs2, &TH.tmr_racs2, &TH.tcr_sacr1, &TH.tms_sacr1,
                        &TH.tcr_sacr2, &TH.tms_sacr2, &TH.tnr_racs1, &TH.tnr_racs2, &TH.tnr_sacr1, &TH.tnr_sacr2};
        for (int i = 0; i < 12; ++i) *s[i] = dalloc(n4s);
        TH.tpi_qcfz = dalloc(NTB_C * 45); TH.tni_qcfz = dalloc(NTB_C * 45);
        TH.tpi_qrfz = dalloc(NTB_R * NTB_R1 * 45); TH.tpg_qrfz = dalloc(NTB_R * NTB_R1 * 45);
        TH.tni_qrfz = dalloc(NTB_R * NTB_R1 * 45); TH.tnr_qrfz = dalloc(NTB_R * NTB_R1 * 45);
        TH.tps_iaus = dalloc(NTB_I * NTB_I1); TH.tni_iaus = dalloc(NTB_I * NTB_I1); TH.tpi_ide = dalloc(NTB_I * NTB_I1);
        TH.t_Efrw = dalloc(NBINS * NBINS); TH.t_Efsw = dalloc(NBINS * NBINS);
    }
    TH.Nt_c = p[0]; TH.TNO = p[1]; TH.am_s = p[2]; TH.rho_g = p[3]; TH.av_s = p[4]; TH.bv_s = p[5]; TH.fv_s = p[6];
    TH.av_g = p[7]; TH.bv_g = p[8]; TH.av_i = p[9]; TH.Ef_si = p[10]; TH.Ef_rs = p[11]; TH.Ef_rg = p[12]; TH.Ef_ri = p[13];
    TH.C_cubes = p[14]; TH.C_sqrd = p[15]; TH.mu_r = p[16]; TH.t_adjust = p[17];
    TH.Ef_rw_l = flags[0]; TH.Ef_sw_l = flags[1];
    TH.am_g = TH_PI2 * TH.rho_g / 6.0f;
    fill_decades(TH.r_c, NTB_C, 1.e-6f); fill_decades(TH.r_i, NTB_I, 1.e-10f); fill_decades(TH.r_r, NTB_R, 1.e-6f);
    fill_decades(TH.r_g, NTB_G, 1.e-5f); fill_decades(TH.r_s, NTB_S, 1.e-5f); fill_decades(TH.N0r_exp, NTB_R1, 1.e6f);
    fill_decades(TH.N0g_exp, NTB_G1, 1.e4f); fill_decades(TH.Nt_i, NTB_I1, 1.0f);

    TH.mu_c = fminf(15.f, (1000.E6f / TH.Nt_c + 2.f));
    TH.Sc3 = powf(TH_Sc, 1.f / 3.f);
    TH.D0i = powf(TH_xm0i / TH_am_i, 1.f / TH_bm_i);
    TH.xm0s = TH.am_s * powf(TH_D0s, TH_bm_s);
    TH.xm0g = TH.am_g * powf(TH_D0g, TH_bm_g);

    float *cce = TH.cce, *ccg = TH.ccg, *cie = TH.cie, *cig = TH.cig, *cre = TH.cre, *crg = TH.crg;
    float *cse = TH.cse, *csg = TH.csg, *cge = TH.cge, *cgg = TH.cgg;
    const float mu_c = TH.mu_c, mu_r = TH.mu_r, bv_s = TH.bv_s, bv_g = TH.bv_g;
    cce[0] = mu_c + 1.f; cce[1] = TH_bm_r + mu_c + 1.f; cce[2] = TH_bm_r + mu_c + 4.f;
    for (int n = 0; n < 3; ++n) ccg[n] = th_wgamma(cce[n]);
    TH.ocg1 = 1.f / ccg[0]; TH.ocg2 = 1.f / ccg[1];
    cie[0] = TH_mu_i + 1.f; cie[1] = TH_bm_i + TH_mu_i + 1.f; cie[2] = TH_bm_i + TH_mu_i + TH_bv_i + 1.f;
    cie[3] = TH_mu_i + TH_bv_i + 1.f; cie[4] = TH_mu_i + 2.f; cie[5] = TH_bm_i * 0.5f + TH_mu_i + TH_bv_i + 1.f;
    cie[6] = TH_bm_i * 0.5f + TH_mu_i + 1.f;
    for (int n = 0; n < 7; ++n) cig[n] = th_wgamma(cie[n]);
    TH.oig1 = 1.f / cig[0]; TH.oig2 = 1.f / cig[1]; TH.obmi = 1.f / TH_bm_i;
    cre[0] = TH_bm_r + 1.f; cre[1] = mu_r + 1.f; cre[2] = TH_bm_r + mu_r + 1.f; cre[3] = TH_bm_r * 2.f + mu_r + 1.f;
    cre[4] = mu_r + TH_bv_r + 1.f; cre[5] = TH_bm_r + mu_r + TH_bv_r + 1.f; cre[6] = TH_bm_r * 0.5f + mu_r + TH_bv_r + 1.f;
    cre[7] = TH_bm_r + mu_r + TH_bv_r + 3.f; cre[8] = mu_r + TH_bv_r + 3.f; cre[9] = mu_r + 2.f;
    cre[10] = 0.5f * (TH_bv_r + 5.f + 2.f * mu_r); cre[11] = TH_bm_r * 0.5f + mu_r + 1.f; cre[12] = TH_bm_r * 2.f + mu_r + TH_bv_r + 1.f;
    for (int n = 0; n < 13; ++n) crg[n] = th_wgamma(cre[n]);
    TH.obmr = 1.f / TH_bm_r; TH.ore1 = 1.f / cre[0]; TH.org1 = 1.f / crg[0]; TH.org2 = 1.f / crg[1]; TH.org3 = 1.f / crg[2];
    cse[0] = TH_bm_s + 1.f; cse[1] = TH_bm_s + 2.f; cse[2] = TH_bm_s * 2.f; cse[3] = TH_bm_s + bv_s + 1.f;
    cse[4] = TH_bm_s * 2.f + bv_s + 1.f; cse[5] = TH_bm_s * 2.f + 1.f; cse[6] = TH_bm_s + TH_mu_s + 1.f;
    cse[7] = TH_bm_s + TH_mu_s + 2.f; cse[8] = TH_bm_s + TH_mu_s + 3.f; cse[9] = TH_bm_s + TH_mu_s + bv_s + 1.f;
    cse[10] = TH_bm_s * 2.f + TH_mu_s + bv_s + 1.f; cse[11] = TH_bm_s * 2.f + TH_mu_s + 1.f; cse[12] = bv_s + 2.f;
    cse[13] = TH_bm_s + bv_s; cse[14] = TH_mu_s + 1.f; cse[15] = 1.0f + (1.0f + bv_s) / 2.f;
    cse[16] = cse[15] + TH_mu_s + 1.f; cse[17] = bv_s + TH_mu_s + 3.f;
    for (int n = 0; n < 18; ++n) csg[n] = th_wgamma(cse[n]);
    TH.oams = 1.f / TH.am_s; TH.obms = 1.f / TH_bm_s; TH.ocms = powf(TH.oams, TH.obms);
    cge[0] = TH_bm_g + 1.f; cge[1] = TH_mu_g + 1.f; cge[2] = TH_bm_g + TH_mu_g + 1.f; cge[3] = TH_bm_g * 2.f + TH_mu_g + 1.f;
    cge[4] = TH_bm_g * 2.f + TH_mu_g + bv_g + 1.f; cge[5] = TH_bm_g + TH_mu_g + bv_g + 1.f; cge[6] = TH_bm_g + TH_mu_g + bv_g + 2.f;
    cge[7] = TH_bm_g + TH_mu_g + bv_g + 3.f; cge[8] = TH_mu_g + bv_g + 3.f; cge[9] = TH_mu_g + 2.f;
    cge[10] = 0.5f * (bv_g + 5.f + 2.f * TH_mu_g); cge[11] = 0.5f * (bv_g + 5.f) + TH_mu_g;
    for (int n = 0; n < 12; ++n) cgg[n] = th_wgamma(cge[n]);
    TH.oamg = 1.f / TH.am_g; TH.obmg = 1.f / TH_bm_g; TH.ocmg = powf(TH.oamg, TH.obmg);
    TH.oge1 = 1.f / cge[0]; TH.ogg1 = 1.f / cgg[0]; TH.ogg2 = 1.f / cgg[1]; TH.ogg3 = 1.f / cgg[2];

    /* rate-equation constants :538-568 */
    TH.t1_qr_qc = TH_PI2 * .25f * TH_av_r * crg[8];
    TH.t1_qr_qi = TH_PI2 * .25f * TH_av_r * crg[8];
    TH.t2_qr_qi = TH_PI2 * .25f * TH_am_r * TH_av_r * crg[7];
    TH.t1_qg_qc = TH_PI2 * .25f * TH.av_g * cgg[8];
    TH.t1_qs_qc = TH_PI2 * .25f * TH.av_s;
    TH.t1_qs_qi = TH_PI2 * .25f * TH.av_s;
    TH.t1_qr_ev = 0.78f * crg[9];
    TH.t2_qr_ev = 0.308f * TH.Sc3 * sqrtf(TH_av_r) * crg[10];
    TH.t1_qs_sd = 0.86f;
    TH.t2_qs_sd = 0.28f * TH.Sc3 * sqrtf(TH.av_s);
    TH.t1_qs_me = TH_PI2 * 4.f * TH.C_sqrd * TH_olfus * 0.86f;
    TH.t2_qs_me = TH_PI2 * 4.f * TH.C_sqrd * TH_olfus * 0.28f * TH.Sc3 * sqrtf(TH.av_s);
    TH.t1_qg_sd = 0.86f * cgg[9];
    TH.t2_qg_sd = 0.28f * TH.Sc3 * sqrtf(TH.av_g) * cgg[10];
    TH.t1_qg_me = TH_PI2 * 4.f * TH_C_cube * TH_olfus * 0.86f * cgg[9];
    TH.t2_qg_me = TH_PI2 * 4.f * TH_C_cube * TH_olfus * 0.28f * TH.Sc3 * sqrtf(TH.av_g) * cgg[10];

    /* table index helpers :571-578 */
    TH.nic2 = (int)lroundf(log10f(TH.r_c[0])); TH.nii2 = (int)lroundf(log10f(TH.r_i[0])); TH.nii3 = (int)lroundf(log10f(TH.Nt_i[0]));
    TH.nir2 = (int)lroundf(log10f(TH.r_r[0])); TH.nir3 = (int)lroundf(log10f(TH.N0r_exp[0])); TH.nis2 = (int)lroundf(log10f(TH.r_s[0]));
    TH.nig2 = (int)lroundf(log10f(TH.r_g[0])); TH.nig3 = (int)lroundf(log10f(TH.N0g_exp[0]));

    /* size bins :581-634 */
    TH.Dc[0] = (double)TH_D0c * 1.0; TH.dtc[0] = (double)TH_D0c * 1.0;
    for (int n = 1; n < NBINS; ++n) { TH.Dc[n] = TH.Dc[n - 1] + 1.0e-6; TH.dtc[n] = TH.Dc[n] - TH.Dc[n - 1]; }
    make_bins((double)TH.D0i * 1.0, 5.0 * (double)TH_D0s, TH.Di, TH.dti);
    make_bins((double)TH_D0r * 1.0, 0.005, TH.Dr, TH.dtr);
    make_bins((double)TH_D0s * 1.0, 0.02, TH.Ds, TH.dts);
    make_bins((double)TH_D0g * 1.0, 0.05, TH.Dg, TH.dtg);

    if (build_tables) {
        table_Efrw(); table_Efsw();
        qr_acr_qg(); qr_acr_qs(); freezeH2O(); qi_aut_qs();
    }
    TH.initialized = 1;
}

struct thompson_state *orc_thompson_state(void) { return &TH; }

/* table access by name for the tests (same names as the reference's module variables) */
const double *orc_thompson_table(const char *name, size_t *n)
{
    const size_t n4g = (size_t)NTB_G1 * NTB_G * NTB_R1 * NTB_R, n4s = (size_t)NTB_S * NTB_T * NTB_R1 * NTB_R;
    const size_t n3 = (size_t)NTB_R * NTB_R1 * 45, n2c = (size_t)NTB_C * 45, n2i = (size_t)NTB_I * NTB_I1, n2e = (size_t)NBINS * NBINS;
#define T(nm, cnt) if (!strcmp(name, #nm)) { *n = cnt; return TH.nm; }
    T(tcg_racg, n4g) T(tmr_racg, n4g) T(tcr_gacr, n4g) T(tmg_gacr, n4g) T(tnr_racg, n4g) T(tnr_gacr, n4g)
    T(tcs_racs1, n4s) T(tmr_racs1, n4s) T(tcs_racs2, n4s) T(tmr_racs2, n4s) T(tcr_sacr1, n4s) T(tms_sacr1, n4s)
    T(tcr_sacr2, n4s) T(tms_sacr2, n4s) T(tnr_racs1, n4s) T(tnr_racs2, n4s) T(tnr_sacr1, n4s) T(tnr_sacr2, n4s)
    T(tpi_qcfz, n2c) T(tni_qcfz, n2c) T(tpi_qrfz, n3) T(tpg_qrfz, n3) T(tni_qrfz, n3) T(tnr_qrfz, n3)
    T(tps_iaus, n2i) T(tni_iaus, n2i) T(tpi_ide, n2i) T(t_Efrw, n2e) T(t_Efsw, n2e)
#undef T
    *n = 0; return NULL;
}
