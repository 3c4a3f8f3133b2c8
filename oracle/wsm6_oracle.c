/* oracle/wsm6_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * WSM6 microphysics (Hong and Lim 2006) on the CPU: a restatement of src/physics/mp_wsm6.f90 -- wsm6 :62-182, wsm62D :185-1384,
 * rgmma :1386-1405, wsm6init :1432-1506, slope_wsm6 / slope_rain / slope_snow / slope_graup :1508-1720, nislfv_rain_plm
 * :1723-1961, nislfv_rain_plm6 :1963-2230 -- written for this checker only (the product's device code in
 * icar_amd/csrc/mp_wsm6.hip is a separate text with a different decomposition).  The reference works on (i,k) slabs of one j
 * row; nothing couples the columns of a slab, so this file walks column by column with every statement of a column in the
 * reference's order and REAL(4) operation order.
 * PINNED by execution: tests/test_oracle_wsm6.py compares it bit-for-bit with the unmodified mp_wsm6.f90 compiled into
 * oracle/_ref (the 60 constants of wsm6init and whole tiles over several steps).
 * Math mode (icar_oracle.c: orc_set_math_mode): 0 = libm expf/logf/powf as the compiled Fortran calls them, 1 = the FP64
 * function rounded once (what the HIP kernels evaluate).
 */
#include <math.h>
#include <stddef.h>
extern int g_math_mode;
static inline float x_exp(float x) { return g_math_mode ? (float)exp((double)x) : expf(x); }
static inline float x_log(float x) { return g_math_mode ? (float)log((double)x) : logf(x); }
static inline float x_pow(float x, float y) { return g_math_mode ? (float)pow((double)x, (double)y) : powf(x, y); }
static inline float fmx(float a, float b) { return a > b ? a : b; }      /* Fortran max / min of two reals */
static inline float fmn(float a, float b) { return a < b ? a : b; }

#define MAXK 128
/* module parameters :16-43 */
static const float dtcldcr = 120.f, n0r = 8.e6f, n0g = 4.e6f, avtr = 841.9f, bvtr = 0.8f, r0 = .8e-5f, peaut = .55f, xncr = 3.e8f,
                   xmyu = 1.718e-5f, avts = 11.72f, bvts = .41f, avtg = 330.f, bvtg = 0.8f, deng = 500.f, n0smax = 1.e11f,
                   lamdarmax = 8.e4f, lamdasmax = 1.e5f, lamdagmax = 6.e4f, dicon = 11.9f, dimax = 500.e-6f, n0s = 2.e6f,
                   alpha = .12f, pfrz1 = 100.f, pfrz2 = 0.66f, qcrmin = 1.e-9f, eacrc = 1.0f, dens = 100.0f, qs0 = 6.e-4f;

/* the SAVE variables wsm6init derives, in the order of their declaration :44-58 */
typedef struct {
    float qc0, qck1, bvtr1, bvtr2, bvtr3, bvtr4, g1pbr, g3pbr, g4pbr, g5pbro2, pvtr, eacrr, pacrr, bvtr6, g6pbr, precr1, precr2, roqimax,
          bvts1, bvts2, bvts3, bvts4, g1pbs, g3pbs, g4pbs, g5pbso2, pvts, pacrs, precs1, precs2, pidn0r, pidn0s, xlv1, pacrc, pi,
          bvtg1, bvtg2, bvtg3, bvtg4, g1pbg, g3pbg, g4pbg, g5pbgo2, pvtg, pacrg, precg1, precg2, pidn0g,
          rslopermax, rslopesmax, rslopegmax, rsloperbmax, rslopesbmax, rslopegbmax, rsloper2max, rslopes2max, rslopeg2max,
          rsloper3max, rslopes3max, rslopeg3max;
} w6_consts;
static w6_consts K;

/* what mp_driver.f90:518-550 passes */
typedef struct { float delt, g, cpd, cpv, rd, rv, t0c, ep1, ep2, qmin, xls, xlv0, xlf0, den0, denr, cliq, cice, psat; } w6_args;

static float rgmma(float x)                                                     /* :1386-1405 */
{
    const float euler = 0.577215664901532f;
    if (x == 1.f) return 0.f;
    float r = x * expf(euler * x);
    for (int i = 1; i <= 10000; ++i) { const float y = (float)i; r = r * (1.000f + x / y) * expf(-x / y); }
    return 1.f / r;
}

/* wsm6init :1432-1506 (host libm, like the compiled reference; once) */
void orc_wsm6_init(float den0, float denr, float dens_, float cl, float cpv, float *out)
{
    (void)dens_;                                             /* the dummy shadows nothing: the module PARAMETER dens is private and unused here */
    K.pi = 4.f * atanf(1.f);
    K.xlv1 = cl - cpv;
    K.qc0 = 4.f / 3.f * K.pi * denr * (r0 * r0 * r0) * xncr / den0;
    K.qck1 = .104f * 9.8f * peaut / powf(xncr * denr, 1.f / 3.f) / xmyu * powf(den0, 4.f / 3.f);
    K.bvtr1 = 1.f + bvtr; K.bvtr2 = 2.5f + .5f * bvtr; K.bvtr3 = 3.f + bvtr; K.bvtr4 = 4.f + bvtr; K.bvtr6 = 6.f + bvtr;
    K.g1pbr = rgmma(K.bvtr1); K.g3pbr = rgmma(K.bvtr3); K.g4pbr = rgmma(K.bvtr4); K.g6pbr = rgmma(K.bvtr6); K.g5pbro2 = rgmma(K.bvtr2);
    K.pvtr = avtr * K.g4pbr / 6.f;
    K.eacrr = 1.0f;
    K.pacrr = K.pi * n0r * avtr * K.g3pbr * .25f * K.eacrr;
    K.precr1 = 2.f * K.pi * n0r * .78f;
    K.precr2 = 2.f * K.pi * n0r * .31f * powf(avtr, .5f) * K.g5pbro2;
    { const float d2 = dimax * dimax, d4 = d2 * d2; K.roqimax = 2.08e22f * (d4 * d4); }
    K.bvts1 = 1.f + bvts; K.bvts2 = 2.5f + .5f * bvts; K.bvts3 = 3.f + bvts; K.bvts4 = 4.f + bvts;
    K.g1pbs = rgmma(K.bvts1); K.g3pbs = rgmma(K.bvts3); K.g4pbs = rgmma(K.bvts4); K.g5pbso2 = rgmma(K.bvts2);
    K.pvts = avts * K.g4pbs / 6.f;
    K.pacrs = K.pi * n0s * avts * K.g3pbs * .25f;
    K.precs1 = 4.f * n0s * .65f;
    K.precs2 = 4.f * n0s * .44f * powf(avts, .5f) * K.g5pbso2;
    K.pidn0r = K.pi * denr * n0r;
    K.pidn0s = K.pi * dens_ * n0s;
    K.pacrc = K.pi * n0s * avts * K.g3pbs * .25f * eacrc;
    K.bvtg1 = 1.f + bvtg; K.bvtg2 = 2.5f + .5f * bvtg; K.bvtg3 = 3.f + bvtg; K.bvtg4 = 4.f + bvtg;
    K.g1pbg = rgmma(K.bvtg1); K.g3pbg = rgmma(K.bvtg3); K.g4pbg = rgmma(K.bvtg4);
    K.pacrg = K.pi * n0g * avtg * K.g3pbg * .25f;
    K.g5pbgo2 = rgmma(K.bvtg2);
    K.pvtg = avtg * K.g4pbg / 6.f;
    K.precg1 = 2.f * K.pi * n0g * .78f;
    K.precg2 = 2.f * K.pi * n0g * .31f * powf(avtg, .5f) * K.g5pbgo2;
    K.pidn0g = K.pi * deng * n0g;
    K.rslopermax = 1.f / lamdarmax; K.rslopesmax = 1.f / lamdasmax; K.rslopegmax = 1.f / lamdagmax;
    K.rsloperbmax = powf(K.rslopermax, bvtr); K.rslopesbmax = powf(K.rslopesmax, bvts); K.rslopegbmax = powf(K.rslopegmax, bvtg);
    K.rsloper2max = K.rslopermax * K.rslopermax; K.rslopes2max = K.rslopesmax * K.rslopesmax; K.rslopeg2max = K.rslopegmax * K.rslopegmax;
    K.rsloper3max = K.rsloper2max * K.rslopermax; K.rslopes3max = K.rslopes2max * K.rslopesmax; K.rslopeg3max = K.rslopeg2max * K.rslopegmax;
    const float *p = (const float *)&K;
    if (out) for (int i = 0; i < (int)(sizeof(w6_consts) / sizeof(float)); ++i) out[i] = p[i];
}

/* ---- slopes: one species of one level (slope_rain / slope_snow / slope_graup, and the three blocks of slope_wsm6) ---- */
static float n0sfac_of(float t) { const float supcol = 273.15f - t; return fmx(fmn(x_exp(alpha * supcol), n0smax / n0s), 1.f); }

static float slope_r(float q, float den, float denfac, float *rs, float *rsb, float *rs2, float *rs3)
{
    if (q <= qcrmin) { *rs = K.rslopermax; *rsb = K.rsloperbmax; *rs2 = K.rsloper2max; *rs3 = K.rsloper3max; }
    else { *rs = 1.f / sqrtf(sqrtf(K.pidn0r / (q * den))); *rsb = x_pow(*rs, bvtr); *rs2 = *rs * *rs; *rs3 = *rs2 * *rs; }
    float vt = K.pvtr * *rsb * denfac;
    if (q <= 0.0f) vt = 0.0f;
    return vt;
}
static float slope_s(float q, float den, float denfac, float t, float *rs, float *rsb, float *rs2, float *rs3)
{
    const float nf = n0sfac_of(t);
    if (q <= qcrmin) { *rs = K.rslopesmax; *rsb = K.rslopesbmax; *rs2 = K.rslopes2max; *rs3 = K.rslopes3max; }
    else { *rs = 1.f / sqrtf(sqrtf(K.pidn0s * nf / (q * den))); *rsb = x_pow(*rs, bvts); *rs2 = *rs * *rs; *rs3 = *rs2 * *rs; }
    float vt = K.pvts * *rsb * denfac;
    if (q <= 0.0f) vt = 0.0f;
    return vt;
}
static float slope_g(float q, float den, float denfac, float *rs, float *rsb, float *rs2, float *rs3)
{
    if (q <= qcrmin) { *rs = K.rslopegmax; *rsb = K.rslopegbmax; *rs2 = K.rslopeg2max; *rs3 = K.rslopeg3max; }
    else { *rs = 1.f / sqrtf(sqrtf(K.pidn0g / (q * den))); *rsb = x_pow(*rs, bvtg); *rs2 = *rs * *rs; *rs3 = *rs2 * *rs; }
    float vt = K.pvtg * *rsb * denfac;
    if (q <= 0.0f) vt = 0.0f;
    return vt;
}

/* ---- semi-Lagrangian fall, shared pieces of nislfv_rain_plm (:1723-1961) and nislfv_rain_plm6 (:1963-2230) ---- */
/* interface fall speeds and arrival points from the cell speeds ww (:1755-1790 / :2003-2038) */
static void fall_arrival(int km, const float *ww, const float *dz, const float *zi, float dt, float *wi, float *za, float *dza)
{
    const float fa1 = 9.f / 16.f, fa2 = 1.f / 16.f, con1 = 0.05f;
    wi[0] = ww[0];
    wi[1] = 0.5f * (ww[1] + ww[0]);
    for (int k = 2; k < km - 1; ++k) wi[k] = fa1 * (ww[k] + ww[k - 1]) - fa2 * (ww[k + 1] + ww[k - 2]);
    wi[km - 1] = 0.5f * (ww[km - 1] + ww[km - 2]);
    wi[km] = ww[km - 1];
    for (int k = 1; k < km; ++k) if (ww[k] == 0.0f) wi[k] = ww[k - 1];
    for (int k = km - 1; k >= 0; --k) {
        const float decfl = (wi[k + 1] - wi[k]) * dt / dz[k];
        if (decfl > con1) wi[k] = wi[k + 1] - con1 * dz[k] / dt;
    }
    for (int k = 0; k <= km; ++k) za[k] = zi[k] - wi[k] * dt;
    for (int k = 0; k < km; ++k) dza[k] = za[k + 1] - za[k];
    dza[km] = zi[km] - za[km];
}

/* piecewise-linear reconstruction, remap onto the regular grid, rain-out (:1815-1948 / :2071-2213); returns precip */
static float fall_remap(int km, const float *zi, const float *za, const float *dza, const float *qa, float *qn)
{
    float qmi[MAXK + 1], qpi[MAXK + 1], precip = 0.f;
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
    for (int k = 0; k < km; ++k) qn[k] = 0.0f;
    int kb = 1, kt = 1;                                      /* 1-based like the reference's */
    for (int k = 1; k <= km; ++k) {
        kb = kb - 1 > 1 ? kb - 1 : 1;
        kt = kt - 1 > 1 ? kt - 1 : 1;
        if (zi[k - 1] >= za[km]) break;
        for (int kk = kb; kk <= km; ++kk) if (zi[k - 1] <= za[kk]) { kb = kk; break; }
        for (int kk = kt; kk <= km; ++kk) if (zi[k] <= za[kk - 1]) { kt = kk; break; }
        kt = kt - 1;
        if (kt == kb) {
            const float tl = (zi[k - 1] - za[kb - 1]) / dza[kb - 1];
            const float th = (zi[k] - za[kb - 1]) / dza[kb - 1];
            const float tl2 = tl * tl, th2 = th * th;
            const float qqd = 0.5f * (qpi[kb - 1] - qmi[kb - 1]);
            const float qqh = qqd * th2 + qmi[kb - 1] * th;
            const float qql = qqd * tl2 + qmi[kb - 1] * tl;
            qn[k - 1] = (qqh - qql) / (th - tl);
        } else if (kt > kb) {
            const float tl = (zi[k - 1] - za[kb - 1]) / dza[kb - 1];
            const float tl2 = tl * tl;
            float qqd = 0.5f * (qpi[kb - 1] - qmi[kb - 1]);
            const float qql = qqd * tl2 + qmi[kb - 1] * tl;
            const float dql = qa[kb - 1] - qql;
            float zsum = (1.f - tl) * dza[kb - 1];
            float qsum = dql * dza[kb - 1];
            if (kt - kb > 1) for (int m = kb + 1; m <= kt - 1; ++m) { zsum = zsum + dza[m - 1]; qsum = qsum + qa[m - 1] * dza[m - 1]; }
            const float th = (zi[k] - za[kt - 1]) / dza[kt - 1];
            const float th2 = th * th;
            qqd = 0.5f * (qpi[kt - 1] - qmi[kt - 1]);
            const float dqh = qqd * th2 + qmi[kt - 1] * th;
            zsum = zsum + th * dza[kt - 1];
            qsum = qsum + dqh * dza[kt - 1];
            qn[k - 1] = qsum / zsum;
        }
    }
    for (int k = 0; k < km; ++k) {
        if (za[k] < 0.0f && za[k + 1] < 0.0f) { precip = precip + qa[k] * dza[k]; continue; }
        else if (za[k] < 0.0f && za[k + 1] >= 0.0f) { precip = precip + qa[k] * (0.0f - za[k]); break; }
        break;
    }
    return precip;
}

/* nislfv_rain_plm for one column: rql = den*q in / out; iter = 1 refines the fall speed once with slope_rain (rain), iter = 0 not (ice) */
static float fall_plm(int km, const float *den, const float *denfac, const float *dz, const float *wwl, float *rql, float dt, int iter)
{
    float ww[MAXK], wi[MAXK + 1], zi[MAXK + 1], za[MAXK + 1], dza[MAXK + 1], qa[MAXK + 1], qn[MAXK];
    float allold = 0.0f;
    for (int k = 0; k < km; ++k) { ww[k] = wwl[k]; allold = allold + rql[k]; }
    if (allold <= 0.0f) return 0.0f;
    zi[0] = 0.0f;
    for (int k = 0; k < km; ++k) zi[k + 1] = zi[k] + dz[k];
    for (int n = 1;; ++n) {
        fall_arrival(km, ww, dz, zi, dt, wi, za, dza);
        for (int k = 0; k < km; ++k) qa[k] = rql[k] * dz[k] / dza[k];
        qa[km] = 0.0f;
        if (n > iter) break;
        for (int k = 0; k < km; ++k) {
            float a, b, c, d;
            const float wa = slope_r(qa[k] / den[k], den[k], denfac[k], &a, &b, &c, &d);
            ww[k] = 0.5f * (wwl[k] + wa);                     /* n == 1: no averaging with the previous estimate */
        }
    }
    const float precip = fall_remap(km, zi, za, dza, qa, qn);
    for (int k = 0; k < km; ++k) rql[k] = qn[k];
    return precip;
}

/* nislfv_rain_plm6 for one column: snow (rql) and graupel (rql2) fall with ONE mass-weighted speed; iter = 1 */
static void fall_plm6(int km, const float *den, const float *denfac, const float *tk, const float *dz, const float *wwl, float *rql, float *rql2,
                      float dt, int iter, float *precip1, float *precip2)
{
    float ww[MAXK], wi[MAXK + 1], zi[MAXK + 1], za[MAXK + 1], dza[MAXK + 1], qa[MAXK + 1], qa2[MAXK + 1], qn[MAXK];
    *precip1 = 0.0f; *precip2 = 0.0f;
    float allold = 0.0f;
    for (int k = 0; k < km; ++k) { ww[k] = wwl[k]; allold = allold + rql[k] + rql2[k]; }
    if (allold <= 0.0f) return;
    zi[0] = 0.0f;
    for (int k = 0; k < km; ++k) zi[k + 1] = zi[k] + dz[k];
    for (int n = 1;; ++n) {
        fall_arrival(km, ww, dz, zi, dt, wi, za, dza);
        for (int k = 0; k < km; ++k) { qa[k] = rql[k] * dz[k] / dza[k]; qa2[k] = rql2[k] * dz[k] / dza[k]; }
        qa[km] = 0.0f; qa2[km] = 0.0f;
        if (n > iter) break;
        for (int k = 0; k < km; ++k) {
            float a, b, c, d;
            const float qr = qa[k] / den[k], qr2 = qa2[k] / den[k];
            float wa = slope_s(qr, den[k], denfac[k], tk[k], &a, &b, &c, &d);
            const float wa2 = slope_g(qr2, den[k], denfac[k], &a, &b, &c, &d);
            const float tmp = fmx(qr + qr2, 1.E-15f);
            if (tmp > 1.e-15f) wa = (wa * qr + wa2 * qr2) / tmp; else wa = 0.f;
            ww[k] = 0.5f * (wwl[k] + wa);
        }
    }
    *precip1 = fall_remap(km, zi, za, dza, qa, qn);
    for (int k = 0; k < km; ++k) rql[k] = qn[k];
    *precip2 = fall_remap(km, zi, za, dza, qa2, qn);
    for (int k = 0; k < km; ++k) rql2[k] = qn[k];
}

/* statement functions of wsm62D :352-366 */
#define CPMCAL(x) (A->cpd * (1.f - fmx(x, A->qmin)) + fmx(x, A->qmin) * A->cpv)
#define XLCAL(x) (A->xlv0 - K.xlv1 * ((x) - A->t0c))
#define DIFFUS(x, y) (8.794e-5f * x_exp(x_log(x) * (1.81f)) / (y))
#define VISCOS(x, y) (1.496e-6f * ((x) * sqrtf(x)) / ((x) + 120.f) / (y))
#define XKA(x, y) (1.414e3f * VISCOS(x, y) * (y))
#define DIFFAC(a, b, c, d, e) ((d) * (a) * (a) / (XKA(c, d) * A->rv * (c) * (c)) + 1.f / ((e) * DIFFUS(c, b)))
#define VENFAC(a, b, c) (x_exp(x_log((VISCOS(b, c) / DIFFUS(b, a))) * ((.3333333f))) / sqrtf(VISCOS(b, c)) * sqrtf(sqrtf(A->den0 / (c))))
#define CONDEN(a, b, c, d, e) ((fmx(b, A->qmin) - (c)) / (1.f + (d) * (d) / (A->rv * (e)) * (c) / ((a) * (a))))

/* saturation mixing ratios over water (qs1) and ice-below-ttp (qs2), the inlined fpvs :451-477, :1330-1356 */
static void sat_mr(const w6_args *A, float t, float p, float *qs1, float *qs2)
{
    const float hsub = A->xls, hvap = A->xlv0, cvap = A->cpv, ttp = A->t0c + 0.01f;
    const float dldt = cvap - A->cliq, xa = -dldt / A->rv, xb = xa + hvap / (A->rv * ttp);
    const float dldti = cvap - A->cice, xai = -dldti / A->rv, xbi = xai + hsub / (A->rv * ttp);
    float tr = ttp / t, v;
    v = A->psat * x_exp(x_log(tr) * (xa)) * x_exp(xb * (1.f - tr));
    v = fmn(v, 0.99f * p);
    v = A->ep2 * v / (p - v);
    *qs1 = fmx(v, A->qmin);
    tr = ttp / t;
    if (t < ttp) v = A->psat * x_exp(x_log(tr) * (xai)) * x_exp(xbi * (1.f - tr));
    else v = A->psat * x_exp(x_log(tr) * (xa)) * x_exp(xb * (1.f - tr));
    v = fmn(v, 0.99f * p);
    v = A->ep2 * v / (p - v);
    *qs2 = fmx(v, A->qmin);
}

/* wsm62D for ONE column (arrays of km levels, k = 0 the lowest).  q, qc, qi, qr, qs_, qg, t in / out; rain, snow, graupel are the
 * caller's accumulators of this column (REAL(4), += like the reference); sr is set to 0 (:408). */
static void wsm6_column(const w6_args *A, int km, float *t, float *q, float *qc, float *qi, float *qr, float *qs_, float *qg,
                        const float *den, const float *p, const float *delz, float *rain, float *snow, float *graupel, float *sr)
{
    float cpm[MAXK], xl[MAXK], denfac[MAXK], qs1[MAXK], qs2[MAXK], rh1[MAXK], rh2[MAXK], xni[MAXK], n0sfac[MAXK];
    float rs[3][MAXK], rsb[3][MAXK], rs2[3][MAXK], rs3[3][MAXK], vt[3][MAXK];
    float workr[MAXK], worka[MAXK], denq1[MAXK], denq2[MAXK], denq3[MAXK], denqci[MAXK], work1c[MAXK];
    float work1a[MAXK], work1b[MAXK], work2[MAXK];
    const float t0c = A->t0c, qmin = A->qmin, xls = A->xls, xlf0 = A->xlf0, denr = A->denr, cliq = A->cliq, pi = K.pi;

    for (int k = 0; k < km; ++k) {                            /* :373-381 */
        qc[k] = fmx(qc[k], 0.0f); qr[k] = fmx(qr[k], 0.0f); qi[k] = fmx(qi[k], 0.0f); qs_[k] = fmx(qs_[k], 0.0f); qg[k] = fmx(qg[k], 0.0f);
    }
    for (int k = 0; k < km; ++k) { cpm[k] = CPMCAL(q[k]); xl[k] = XLCAL(t[k]); }      /* :388-393 */
    *sr = 0.f;                                                /* :408 */
    /* minor time steps :416-418 */
    const long lp = lroundf(A->delt / dtcldcr);
    const int loops = lp > 1 ? (int)lp : 1;
    float dtcld = A->delt / (float)loops;
    if (A->delt <= dtcldcr) dtcld = A->delt;

    for (int loop = 1; loop <= loops; ++loop) {
        const float mstep = 1.f;                              /* mstep(i) = 1 :426 (an INTEGER there: /mstep is a REAL division by 1.) */
        for (int k = 0; k < km; ++k) {                        /* :438-446 */
            float tv = 1.f / den[k];
            tv = tv * A->den0;
            denfac[k] = sqrtf(tv);
        }
        for (int k = 0; k < km; ++k) {                        /* :451-477 */
            sat_mr(A, t[k], p[k], &qs1[k], &qs2[k]);
            rh1[k] = fmx(q[k] / qs1[k], qmin);
            rh2[k] = fmx(q[k] / qs2[k], qmin);
        }
        /* process rates start at zero :483-529 */
        for (int k = 0; k < km; ++k) {                        /* Ni :534-540 */
            float temp = (den[k] * fmx(qi[k], qmin));
            temp = sqrtf(sqrtf(temp * temp * temp));
            xni[k] = fmn(fmx(5.38e7f * temp, 1.e3f), 1.e6f);
        }
        /* ---- fall of rain, and of snow + graupel :546-589 ---- */
        for (int k = 0; k < km; ++k) {
            vt[0][k] = slope_r(qr[k], den[k], denfac[k], &rs[0][k], &rsb[0][k], &rs2[0][k], &rs3[0][k]);
            vt[1][k] = slope_s(qs_[k], den[k], denfac[k], t[k], &rs[1][k], &rsb[1][k], &rs2[1][k], &rs3[1][k]);
            vt[2][k] = slope_g(qg[k], den[k], denfac[k], &rs[2][k], &rsb[2][k], &rs2[2][k], &rs3[2][k]);
        }
        for (int k = km - 1; k >= 0; --k) {
            workr[k] = vt[0][k];
            const float qsum = fmx((qs_[k] + qg[k]), 1.E-15f);
            if (qsum > 1.e-15f) worka[k] = (vt[1][k] * qs_[k] + vt[2][k] * qg[k]) / qsum; else worka[k] = 0.f;
            denq1[k] = den[k] * qr[k]; denq2[k] = den[k] * qs_[k]; denq3[k] = den[k] * qg[k];
            if (qr[k] <= 0.0f) workr[k] = 0.0f;
        }
        float delqrs1, delqrs2, delqrs3;
        delqrs1 = fall_plm(km, den, denfac, delz, workr, denq1, dtcld, 1);
        fall_plm6(km, den, denfac, t, delz, worka, denq2, denq3, dtcld, 1, &delqrs2, &delqrs3);
        for (int k = 0; k < km; ++k) { qr[k] = fmx(denq1[k] / den[k], 0.f); qs_[k] = fmx(denq2[k] / den[k], 0.f); qg[k] = fmx(denq3[k] / den[k], 0.f); }
        /* only the lowest level of fall(:,:,1:3) is read later (:693-695): :586-588 */
        const float fall1 = delqrs1 / delz[0] / dtcld, fall2 = delqrs2 / delz[0] / dtcld, fall3 = delqrs3 / delz[0] / dtcld;
        for (int k = 0; k < km; ++k) {                        /* :596-597 */
            vt[0][k] = slope_r(qr[k], den[k], denfac[k], &rs[0][k], &rsb[0][k], &rs2[0][k], &rs3[0][k]);
            vt[1][k] = slope_s(qs_[k], den[k], denfac[k], t[k], &rs[1][k], &rsb[1][k], &rs2[1][k], &rs3[1][k]);
            vt[2][k] = slope_g(qg[k], den[k], denfac[k], &rs[2][k], &rsb[2][k], &rs2[2][k], &rs3[2][k]);
        }
        for (int k = km - 1; k >= 0; --k) {                   /* melting of snow and graupel :599-637 */
            const float supcol = t0c - t[k];
            n0sfac[k] = fmx(fmn(x_exp(alpha * supcol), n0smax / n0s), 1.f);
            if (t[k] > t0c) {
                const float xlf = xlf0;
                work2[k] = VENFAC(p[k], t[k], den[k]);
                if (qs_[k] > 0.f) {
                    const float coeres = rs2[1][k] * sqrtf(rs[1][k] * rsb[1][k]);
                    float psmlt = XKA(t[k], den[k]) / xlf * (t0c - t[k]) * pi / 2.f * n0sfac[k] * (K.precs1 * rs2[1][k] + K.precs2 * work2[k] * coeres);
                    psmlt = fmn(fmx(psmlt * dtcld / mstep, -qs_[k] / mstep), 0.f);
                    qs_[k] = qs_[k] + psmlt;
                    qr[k] = qr[k] - psmlt;
                    t[k] = t[k] + xlf / cpm[k] * psmlt;
                }
                if (qg[k] > 0.f) {
                    const float coeres = rs2[2][k] * sqrtf(rs[2][k] * rsb[2][k]);
                    float pgmlt = XKA(t[k], den[k]) / xlf * (t0c - t[k]) * (K.precg1 * rs2[2][k] + K.precg2 * work2[k] * coeres);
                    pgmlt = fmn(fmx(pgmlt * dtcld / mstep, -qg[k] / mstep), 0.f);
                    qg[k] = qg[k] + pgmlt;
                    qr[k] = qr[k] - pgmlt;
                    t[k] = t[k] + xlf / cpm[k] * pgmlt;
                }
            }
        }
        /* ---- fall of cloud ice :641-667 ---- */
        for (int k = km - 1; k >= 0; --k) {
            if (qi[k] <= 0.f) work1c[k] = 0.f;
            else {
                const float xmi = den[k] * qi[k] / xni[k];
                const float diameter = fmx(fmn(dicon * sqrtf(xmi), dimax), 1.e-25f);
                work1c[k] = 1.49e4f * x_exp(x_log(diameter) * (1.31f));
            }
        }
        for (int k = km - 1; k >= 0; --k) denqci[k] = den[k] * qi[k];
        const float delqi = fall_plm(km, den, denfac, delz, work1c, denqci, dtcld, 0);
        for (int k = 0; k < km; ++k) qi[k] = fmx(denqci[k] / den[k], 0.f);
        const float fallc = delqi / delz[0] / dtcld;
        /* surface :672-697 (snowncv / graupelncv are not PRESENT; tstepsnow / tstepgraup feed nothing) */
        {
            const float fallsum = fall1 + fall2 + fall3 + fallc;
            const float fallsum_qsi = fall2 + fallc;
            const float fallsum_qg = fall3;
            if (fallsum > 0.f) *rain = fallsum * delz[0] / denr * dtcld * 1000.f + *rain;
            if (fallsum_qsi > 0.f) *snow = fallsum_qsi * delz[0] / denr * dtcld * 1000.f + *snow;
            if (fallsum_qg > 0.f) *graupel = fallsum_qg * delz[0] / denr * dtcld * 1000.f + *graupel;
        }
        /* pimlt, pihmf, pihtf, pgfrz :703-759 */
        for (int k = 0; k < km; ++k) {
            const float supcol = t0c - t[k];
            float xlf = xls - xl[k];
            if (supcol < 0.f) xlf = xlf0;
            if (supcol < 0.f && qi[k] > 0.f) {
                qc[k] = qc[k] + qi[k];
                t[k] = t[k] - xlf / cpm[k] * qi[k];
                qi[k] = 0.f;
            }
            if (supcol > 40.f && qc[k] > 0.f) {
                qi[k] = qi[k] + qc[k];
                t[k] = t[k] + xlf / cpm[k] * qc[k];
                qc[k] = 0.f;
            }
            if (supcol > 0.f && qc[k] > qmin) {
                const float supcolt = fmn(supcol, 50.f);
                const float pfrzdtc = fmn(pfrz1 * (x_exp(pfrz2 * supcolt) - 1.f) * den[k] / denr / xncr * qc[k] * qc[k] * dtcld, qc[k]);
                qi[k] = qi[k] + pfrzdtc;
                t[k] = t[k] + xlf / cpm[k] * pfrzdtc;
                qc[k] = qc[k] - pfrzdtc;
            }
            if (supcol > 0.f && qr[k] > 0.f) {
                float temp = rs3[0][k];
                temp = temp * temp * rs[0][k];
                const float supcolt = fmn(supcol, 50.f);
                const float pfrzdtr = fmn(20.f * (pi * pi) * pfrz1 * n0r * denr / den[k] * (x_exp(pfrz2 * supcolt) - 1.f) * temp * dtcld, qr[k]);
                qg[k] = qg[k] + pfrzdtr;
                t[k] = t[k] + xlf / cpm[k] * pfrzdtr;
                qr[k] = qr[k] - pfrzdtr;
            }
        }
        /* slopes for the process rates :765-773 */
        for (int k = 0; k < km; ++k) {
            vt[0][k] = slope_r(qr[k], den[k], denfac[k], &rs[0][k], &rsb[0][k], &rs2[0][k], &rs3[0][k]);
            vt[1][k] = slope_s(qs_[k], den[k], denfac[k], t[k], &rs[1][k], &rsb[1][k], &rs2[1][k], &rs3[1][k]);
            vt[2][k] = slope_g(qg[k], den[k], denfac[k], &rs[2][k], &rsb[2][k], &rs2[2][k], &rs3[2][k]);
        }
        for (int k = 0; k < km; ++k) {                        /* :782-789 */
            work1a[k] = DIFFAC(xl[k], p[k], t[k], den[k], qs1[k]);
            work1b[k] = DIFFAC(xls, p[k], t[k], den[k], qs2[k]);
            work2[k] = VENFAC(p[k], t[k], den[k]);
        }
        for (int k = 0; k < km; ++k) {
            float prevp = 0.f, psdep = 0.f, pgdep = 0.f, praut = 0.f, psaut = 0.f, pgaut = 0.f, pracw = 0.f, praci = 0.f, piacr = 0.f, psaci = 0.f,
                  psacw = 0.f, pracs = 0.f, psacr = 0.f, pgacw = 0.f, paacw = 0.f, pgaci = 0.f, pgacr = 0.f, pgacs = 0.f, pigen = 0.f, pidep = 0.f,
                  pseml = 0.f, pgeml = 0.f, psevp = 0.f, pgevp = 0.f;
            /* ---- warm rain :802-840 ---- */
            {
                const float supsat = fmx(q[k], qmin) - qs1[k];
                const float satdt = supsat / dtcld;
                if (qc[k] > K.qc0) {
                    praut = K.qck1 * x_pow(qc[k], 7.f / 3.f);
                    praut = fmn(praut, qc[k] / dtcld);
                }
                if (qr[k] > qcrmin && qc[k] > qmin)
                    pracw = fmn(K.pacrr * rs3[0][k] * rsb[0][k] * qc[k] * denfac[k], qc[k] / dtcld);
                if (qr[k] > 0.f) {
                    const float coeres = rs2[0][k] * sqrtf(rs[0][k] * rsb[0][k]);
                    prevp = (rh1[k] - 1.f) * (K.precr1 * rs2[0][k] + K.precr2 * work2[k] * coeres) / work1a[k];
                    if (prevp < 0.f) {
                        prevp = fmx(prevp, -qr[k] / dtcld);
                        prevp = fmx(prevp, satdt / 2);
                    } else prevp = fmn(prevp, satdt / 2);
                }
            }
            /* ---- cold rain :855-1128 ---- */
            const float supcol = t0c - t[k];
            n0sfac[k] = fmx(fmn(x_exp(alpha * supcol), n0smax / n0s), 1.f);
            const float supsat = fmx(q[k], qmin) - qs2[k];
            const float satdt = supsat / dtcld;
            int ifsat = 0;
            {
                float temp = (den[k] * fmx(qi[k], qmin));
                temp = sqrtf(sqrtf(temp * temp * temp));
                xni[k] = fmn(fmx(5.38e7f * temp, 1.e3f), 1.e6f);
            }
            const float eacrs = x_exp(0.07f * (-supcol));
            const float xmi = den[k] * qi[k] / xni[k];
            const float diameter = fmn(dicon * sqrtf(xmi), dimax);
            const float vt2i = 1.49e4f * x_pow(diameter, 1.31f);
            const float vt2r = K.pvtr * rsb[0][k] * denfac[k];
            const float vt2s = K.pvts * rsb[1][k] * denfac[k];
            const float vt2g = K.pvtg * rsb[2][k] * denfac[k];
            const float qsum = fmx((qs_[k] + qg[k]), 1.E-15f);
            float vt2ave;
            if (qsum > 1.e-15f) vt2ave = (vt2s * qs_[k] + vt2g * qg[k]) / (qsum); else vt2ave = 0.f;
            if (supcol > 0.f && qi[k] > qmin) {
                if (qr[k] > qcrmin) {
                    const float acrfac = 2.f * rs3[0][k] + 2.f * diameter * rs2[0][k] + diameter * diameter * rs[0][k];
                    praci = pi * qi[k] * n0r * fabsf(vt2r - vt2i) * acrfac / 4.f;
                    praci = fmn(praci, qi[k] / dtcld);
                    piacr = pi * pi * avtr * n0r * denr * xni[k] * denfac[k] * K.g6pbr * rs3[0][k] * rs3[0][k] * rsb[0][k] / 24.f / den[k];
                    piacr = fmn(piacr, qr[k] / dtcld);
                }
                if (qs_[k] > qcrmin) {
                    const float acrfac = 2.f * rs3[1][k] + 2.f * diameter * rs2[1][k] + diameter * diameter * rs[1][k];
                    psaci = pi * qi[k] * eacrs * n0s * n0sfac[k] * fabsf(vt2ave - vt2i) * acrfac / 4.f;
                    psaci = fmn(psaci, qi[k] / dtcld);
                }
                if (qg[k] > qcrmin) {
                    const float egi = x_exp(0.07f * (-supcol));
                    const float acrfac = 2.f * rs3[2][k] + 2.f * diameter * rs2[2][k] + diameter * diameter * rs[2][k];
                    pgaci = pi * egi * qi[k] * n0g * fabsf(vt2ave - vt2i) * acrfac / 4.f;
                    pgaci = fmn(pgaci, qi[k] / dtcld);
                }
            }
            if (qs_[k] > qcrmin && qc[k] > qmin)
                psacw = fmn(K.pacrc * n0sfac[k] * rs3[1][k] * rsb[1][k] * qc[k] * denfac[k], qc[k] / dtcld);
            if (qg[k] > qcrmin && qc[k] > qmin)
                pgacw = fmn(K.pacrg * rs3[2][k] * rsb[2][k] * qc[k] * denfac[k], qc[k] / dtcld);
            if (qsum > 1.e-15f) paacw = (qs_[k] * psacw + qg[k] * pgacw) / (qsum);
            if (qs_[k] > qcrmin && qr[k] > qcrmin) {
                if (supcol > 0) {
                    const float acrfac = 5.f * rs3[1][k] * rs3[1][k] * rs[0][k] + 2.f * rs3[1][k] * rs2[1][k] * rs2[0][k] + .5f * rs2[1][k] * rs2[1][k] * rs3[0][k];
                    pracs = pi * pi * n0r * n0s * n0sfac[k] * fabsf(vt2r - vt2ave) * (dens / den[k]) * acrfac;
                    pracs = fmn(pracs, qs_[k] / dtcld);
                }
                const float acrfac = 5.f * rs3[0][k] * rs3[0][k] * rs[1][k] + 2.f * rs3[0][k] * rs2[0][k] * rs2[1][k] + .5f * rs2[0][k] * rs2[0][k] * rs3[1][k];
                psacr = pi * pi * n0r * n0s * n0sfac[k] * fabsf(vt2ave - vt2r) * (denr / den[k]) * acrfac;
                psacr = fmn(psacr, qr[k] / dtcld);
            }
            if (qg[k] > qcrmin && qr[k] > qcrmin) {
                const float acrfac = 5.f * rs3[0][k] * rs3[0][k] * rs[2][k] + 2.f * rs3[0][k] * rs2[0][k] * rs2[2][k] + .5f * rs2[0][k] * rs2[0][k] * rs3[2][k];
                pgacr = pi * pi * n0r * n0g * fabsf(vt2ave - vt2r) * (denr / den[k]) * acrfac;
                pgacr = fmn(pgacr, qr[k] / dtcld);
            }
            if (qg[k] > qcrmin && qs_[k] > qcrmin) pgacs = 0.f;
            if (supcol <= 0) {
                const float xlf = xlf0;
                if (qs_[k] > 0.f) pseml = fmn(fmx(cliq * supcol * (paacw + psacr) / xlf, -qs_[k] / dtcld), 0.f);
                if (qg[k] > 0.f) pgeml = fmn(fmx(cliq * supcol * (paacw + pgacr) / xlf, -qg[k] / dtcld), 0.f);
            }
            if (supcol > 0) {
                if (qi[k] > 0 && ifsat != 1) {
                    pidep = 4.f * diameter * xni[k] * (rh2[k] - 1.f) / work1b[k];
                    const float supice = satdt - prevp;
                    if (pidep < 0.f) {
                        pidep = fmx(fmx(pidep, satdt / 2), supice);
                        pidep = fmx(pidep, -qi[k] / dtcld);
                    } else pidep = fmn(fmn(pidep, satdt / 2), supice);
                    if (fabsf(prevp + pidep) >= fabsf(satdt)) ifsat = 1;
                }
                if (qs_[k] > 0.f && ifsat != 1) {
                    const float coeres = rs2[1][k] * sqrtf(rs[1][k] * rsb[1][k]);
                    psdep = (rh2[k] - 1.f) * n0sfac[k] * (K.precs1 * rs2[1][k] + K.precs2 * work2[k] * coeres) / work1b[k];
                    const float supice = satdt - prevp - pidep;
                    if (psdep < 0.f) {
                        psdep = fmx(psdep, -qs_[k] / dtcld);
                        psdep = fmx(fmx(psdep, satdt / 2), supice);
                    } else psdep = fmn(fmn(psdep, satdt / 2), supice);
                    if (fabsf(prevp + pidep + psdep) >= fabsf(satdt)) ifsat = 1;
                }
                if (qg[k] > 0.f && ifsat != 1) {
                    const float coeres = rs2[2][k] * sqrtf(rs[2][k] * rsb[2][k]);
                    pgdep = (rh2[k] - 1.f) * (K.precg1 * rs2[2][k] + K.precg2 * work2[k] * coeres) / work1b[k];
                    const float supice = satdt - prevp - pidep - psdep;
                    if (pgdep < 0.f) {
                        pgdep = fmx(pgdep, -qg[k] / dtcld);
                        pgdep = fmx(fmx(pgdep, satdt / 2), supice);
                    } else pgdep = fmn(fmn(pgdep, satdt / 2), supice);
                    if (fabsf(prevp + pidep + psdep + pgdep) >= fabsf(satdt)) ifsat = 1;
                }
                if (supsat > 0 && ifsat != 1) {
                    const float supice = satdt - prevp - pidep - psdep - pgdep;
                    const float xni0 = 1.e3f * x_exp(0.1f * supcol);
                    const float roqi0 = 4.92e-11f * x_pow(xni0, 1.33f);
                    pigen = fmx(0.f, (roqi0 / den[k] - fmx(qi[k], 0.f)) / dtcld);
                    pigen = fmn(fmn(pigen, satdt), supice);
                }
                if (qi[k] > 0.f) {
                    const float qimax = K.roqimax / den[k];
                    psaut = fmx(0.f, (qi[k] - qimax) / dtcld);
                }
                if (qs_[k] > 0.f) {
                    const float alpha2 = 1.e-3f * x_exp(0.09f * (-supcol));
                    pgaut = fmn(fmx(0.f, alpha2 * (qs_[k] - qs0)), qs_[k] / dtcld);
                }
            }
            if (supcol < 0.f) {
                if (qs_[k] > 0.f && rh1[k] < 1.f) {
                    const float coeres = rs2[1][k] * sqrtf(rs[1][k] * rsb[1][k]);
                    psevp = (rh1[k] - 1.f) * n0sfac[k] * (K.precs1 * rs2[1][k] + K.precs2 * work2[k] * coeres) / work1a[k];
                    psevp = fmn(fmx(psevp, -qs_[k] / dtcld), 0.f);
                }
                if (qg[k] > 0.f && rh1[k] < 1.f) {
                    const float coeres = rs2[2][k] * sqrtf(rs[2][k] * rsb[2][k]);
                    pgevp = (rh1[k] - 1.f) * (K.precg1 * rs2[2][k] + K.precg2 * work2[k] * coeres) / work1a[k];
                    pgevp = fmn(fmx(pgevp, -qg[k] / dtcld), 0.f);
                }
            }
            /* ---- conservation and update :1136-1318 ---- */
            float delta2 = 0.f, delta3 = 0.f, value, source, factor;
            if (qr[k] < 1.e-4f && qs_[k] < 1.e-4f) delta2 = 1.f;
            if (qr[k] < 1.e-4f) delta3 = 1.f;
            if (t[k] <= t0c) {
                value = fmx(qmin, qc[k]);
                source = (praut + pracw + paacw + paacw) * dtcld;
                if (source > value) { factor = value / source; praut = praut * factor; pracw = pracw * factor; paacw = paacw * factor; }
                value = fmx(qmin, qi[k]);
                source = (psaut - pigen - pidep + praci + psaci + pgaci) * dtcld;
                if (source > value) {
                    factor = value / source;
                    psaut = psaut * factor; pigen = pigen * factor; pidep = pidep * factor; praci = praci * factor; psaci = psaci * factor; pgaci = pgaci * factor;
                }
                value = fmx(qmin, qr[k]);
                source = (-praut - prevp - pracw + piacr + psacr + pgacr) * dtcld;
                if (source > value) {
                    factor = value / source;
                    praut = praut * factor; prevp = prevp * factor; pracw = pracw * factor; piacr = piacr * factor; psacr = psacr * factor; pgacr = pgacr * factor;
                }
                value = fmx(qmin, qs_[k]);
                source = -(psdep + psaut - pgaut + paacw + piacr * delta3 + praci * delta3 - pracs * (1.f - delta2) + psacr * delta2 + psaci - pgacs) * dtcld;
                if (source > value) {
                    factor = value / source;
                    psdep = psdep * factor; psaut = psaut * factor; pgaut = pgaut * factor; paacw = paacw * factor; piacr = piacr * factor;
                    praci = praci * factor; psaci = psaci * factor; pracs = pracs * factor; psacr = psacr * factor; pgacs = pgacs * factor;
                }
                value = fmx(qmin, qg[k]);
                source = -(pgdep + pgaut + piacr * (1.f - delta3) + praci * (1.f - delta3) + psacr * (1.f - delta2) + pracs * (1.f - delta2)
                           + pgaci + paacw + pgacr + pgacs) * dtcld;
                if (source > value) {
                    factor = value / source;
                    pgdep = pgdep * factor; pgaut = pgaut * factor; piacr = piacr * factor; praci = praci * factor; psacr = psacr * factor;
                    pracs = pracs * factor; paacw = paacw * factor; pgaci = pgaci * factor; pgacr = pgacr * factor; pgacs = pgacs * factor;
                }
                const float w2 = -(prevp + psdep + pgdep + pigen + pidep);
                q[k] = q[k] + w2 * dtcld;
                qc[k] = fmx(qc[k] - (praut + pracw + paacw + paacw) * dtcld, 0.f);
                qr[k] = fmx(qr[k] + (praut + pracw + prevp - piacr - pgacr - psacr) * dtcld, 0.f);
                qi[k] = fmx(qi[k] - (psaut + praci + psaci + pgaci - pigen - pidep) * dtcld, 0.f);
                qs_[k] = fmx(qs_[k] + (psdep + psaut + paacw - pgaut + piacr * delta3 + praci * delta3 + psaci - pgacs - pracs * (1.f - delta2)
                                       + psacr * delta2) * dtcld, 0.f);
                qg[k] = fmx(qg[k] + (pgdep + pgaut + piacr * (1.f - delta3) + praci * (1.f - delta3) + psacr * (1.f - delta2) + pracs * (1.f - delta2)
                                     + pgaci + paacw + pgacr + pgacs) * dtcld, 0.f);
                const float xlf = xls - xl[k];
                const float xlwork2 = -xls * (psdep + pgdep + pidep + pigen) - xl[k] * prevp - xlf * (piacr + paacw + paacw + pgacr + psacr);
                t[k] = t[k] - xlwork2 / cpm[k] * dtcld;
            } else {
                value = fmx(qmin, qc[k]);
                source = (praut + pracw + paacw + paacw) * dtcld;
                if (source > value) { factor = value / source; praut = praut * factor; pracw = pracw * factor; paacw = paacw * factor; }
                value = fmx(qmin, qr[k]);
                source = (-paacw - praut + pseml + pgeml - pracw - paacw - prevp) * dtcld;
                if (source > value) {
                    factor = value / source;
                    praut = praut * factor; prevp = prevp * factor; pracw = pracw * factor; paacw = paacw * factor; pseml = pseml * factor; pgeml = pgeml * factor;
                }
                value = fmx(qcrmin, qs_[k]);
                source = (pgacs - pseml - psevp) * dtcld;
                if (source > value) { factor = value / source; pgacs = pgacs * factor; psevp = psevp * factor; pseml = pseml * factor; }
                value = fmx(qcrmin, qg[k]);
                source = -(pgacs + pgevp + pgeml) * dtcld;
                if (source > value) { factor = value / source; pgacs = pgacs * factor; pgevp = pgevp * factor; pgeml = pgeml * factor; }
                const float w2 = -(prevp + psevp + pgevp);
                q[k] = q[k] + w2 * dtcld;
                qc[k] = fmx(qc[k] - (praut + pracw + paacw + paacw) * dtcld, 0.f);
                qr[k] = fmx(qr[k] + (praut + pracw + prevp + paacw + paacw - pseml - pgeml) * dtcld, 0.f);
                qs_[k] = fmx(qs_[k] + (psevp - pgacs + pseml) * dtcld, 0.f);
                qg[k] = fmx(qg[k] + (pgacs + pgevp + pgeml) * dtcld, 0.f);
                const float xlf = xls - xl[k];
                const float xlwork2 = -xl[k] * (prevp + psevp + pgevp) - xlf * (pseml + pgeml);
                t[k] = t[k] - xlwork2 / cpm[k] * dtcld;
            }
        }
        for (int k = 0; k < km; ++k) sat_mr(A, t[k], p[k], &qs1[k], &qs2[k]);            /* :1330-1356 */
        for (int k = 0; k < km; ++k) {                        /* pcond :1364-1374 */
            const float w1 = CONDEN(t[k], q[k], qs1[k], xl[k], cpm[k]);
            float pcond = fmn(fmx(w1 / dtcld, 0.f), fmx(q[k], 0.f) / dtcld);
            if (qc[k] > 0.f && w1 < 0.f) pcond = fmx(w1, -qc[k]) / dtcld;
            q[k] = q[k] - pcond * dtcld;
            qc[k] = fmx(qc[k] + pcond * dtcld, 0.f);
            t[k] = t[k] + pcond * xl[k] / cpm[k] * dtcld;
        }
        for (int k = 0; k < km; ++k) {                        /* :1380-1385 */
            if (qc[k] <= qmin) qc[k] = 0.0f;
            if (qi[k] <= qmin) qi[k] = 0.0f;
        }
    }
}

/* wsm6 (:62-182): t = th*pii, wsm62D per row, th = t/pii.  Arrays X(i,k,j) -> i + nx*(k + nz*j), 1-based inclusive tile bounds.
 * rainncv is an argument the reference never writes (its lines are commented out); sr is zeroed. */
int orc_wsm6(int nx, int nz, int ny, float *th, float *q, float *qc, float *qr, float *qi, float *qs, float *qg, const float *den, const float *pii,
             const float *p, const float *delz, const float *args18, float *rain, float *sr, float *snow, float *graupel,
             int its, int ite, int jts, int jte, int kts, int kte)
{
    w6_args A;
    const int km = kte - kts + 1;
    if (km > MAXK || km < 4) return 1;
    { float *a = (float *)&A; for (int i = 0; i < 18; ++i) a[i] = args18[i]; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = jts - 1; j <= jte - 1; ++j) for (int i = its - 1; i <= ite - 1; ++i) {
        float t[MAXK], cq[MAXK], cqc[MAXK], cqi[MAXK], cqr[MAXK], cqs[MAXK], cqg[MAXK], cden[MAXK], cp[MAXK], cdz[MAXK];
        for (int k = 0; k < km; ++k) {
            const size_t c = (size_t)i + (size_t)nx * ((size_t)(k + kts - 1) + (size_t)nz * j);
            t[k] = th[c] * pii[c]; cq[k] = q[c]; cqc[k] = qc[c]; cqi[k] = qi[c]; cqr[k] = qr[c]; cqs[k] = qs[c]; cqg[k] = qg[c];
            cden[k] = den[c]; cp[k] = p[c]; cdz[k] = delz[c];
        }
        const size_t o = (size_t)i + (size_t)nx * j;
        wsm6_column(&A, km, t, cq, cqc, cqi, cqr, cqs, cqg, cden, cp, cdz, &rain[o], &snow[o], &graupel[o], &sr[o]);
        for (int k = 0; k < km; ++k) {
            const size_t c = (size_t)i + (size_t)nx * ((size_t)(k + kts - 1) + (size_t)nz * j);
            th[c] = t[k] / pii[c]; q[c] = cq[k]; qc[c] = cqc[k]; qi[c] = cqi[k]; qr[c] = cqr[k]; qs[c] = cqs[k]; qg[c] = cqg[k];
        }
    }
    return 0;
}
