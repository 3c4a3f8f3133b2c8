/* oracle/wsm3_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * WSM3 microphysics (Hong, Dudhia and Chen 2004) on the CPU: a restatement of src/physics/mp_wsm3.f90 -- wsm3 :74-216,
 * wsm32D :218-903, rgmma :905-924, wsm3init :951-1006, slope_wsm3 :1008-1068, nislfv_rain_plm :1266-1504 -- written for this
 * checker only, from the Fortran.  It keeps the reference's own decomposition: one (i,k) SLAB per j row, every loop nest of
 * wsm32D a loop nest over the slab here, in the reference's order and REAL(4) operation order.  The product's device code
 * (icar_amd/csrc/wsm3_column.h + mp_wsm3.hip) is a separate text with a different decomposition (one column per lane, the
 * levels marched in registers), so HIP == oracle is a comparison of two independent restatements.
 * PINNED by execution: tests/test_oracle_wsm3.py compares this file bit-for-bit with the unmodified mp_wsm3.f90 compiled into
 * oracle/_ref (the 42 constants of wsm3init and whole tiles over several steps).
 * Math mode (icar_oracle.c: orc_set_math_mode): 0 = libm expf/logf/powf as the compiled Fortran calls them (what the HIP
 * kernels reproduce bit for bit), 1 = the FP64 function rounded once (a sensitivity probe only).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
extern int g_math_mode;
static inline float x_exp(float x) { return g_math_mode ? (float)exp((double)x) : expf(x); }
static inline float x_log(float x) { return g_math_mode ? (float)log((double)x) : logf(x); }
static inline float x_pow(float x, float y) { return g_math_mode ? (float)pow((double)x, (double)y) : powf(x, y); }
static inline float fmx(float a, float b) { return a > b ? a : b; }      /* Fortran max / min of two reals */
static inline float fmn(float a, float b) { return a < b ? a : b; }

/* module parameters :33-52 */
static const float dtcldcr = 120.f, n0r = 8.e6f, avtr = 841.9f, bvtr = 0.8f, r0 = .8e-5f, peaut = .55f, xncr = 3.e8f, xmyu = 1.718e-5f,
                   avts = 11.72f, bvts = .41f, n0smax = 1.e11f, lamdarmax = 8.e4f, lamdasmax = 1.e5f, dicon = 11.9f, dimax = 500.e-6f,
                   n0s = 2.e6f, alpha = .12f, qcrmin = 1.e-9f;

/* the SAVE variables of :53-68 that wsm3init sets, in the order of their declaration */
static struct {
    float qc0, qck1, pidnc, bvtr1, bvtr2, bvtr3, bvtr4, g1pbr, g3pbr, g4pbr, g5pbro2, pvtr, eacrr, pacrr, precr1, precr2, xmmax, roqimax,
          bvts1, bvts2, bvts3, bvts4, g1pbs, g3pbs, g4pbs, g5pbso2, pvts, pacrs, precs1, precs2, pidn0r, pidn0s, xlv1, pi,
          rslopermax, rslopesmax, rsloperbmax, rslopesbmax, rsloper2max, rslopes2max, rsloper3max, rslopes3max;
} S;

static float rgmma(float x)                                                     /* :905-924 */
{
    const float euler = 0.577215664901532f;
    float g, y;
    if (x == 1.f) return 0.f;
    g = x * expf(euler * x);
    for (int i = 1; i <= 10000; ++i) { y = (float)i; g = g * (1.000f + x / y) * expf(-x / y); }
    return 1.f / g;
}

/* wsm3init :951-1006 (host libm like the compiled reference; once).  out[0..41] in the order of S */
void orc_wsm3_init(float den0, float denr, float dens, float cl, float cpv, float *out)
{
    S.pi = 4.f * atanf(1.f);
    S.xlv1 = cl - cpv;
    S.qc0 = 4.f / 3.f * S.pi * denr * (r0 * r0 * r0) * xncr / den0;
    S.qck1 = .104f * 9.8f * peaut / powf(xncr * denr, 1.f / 3.f) / xmyu * powf(den0, 4.f / 3.f);
    S.pidnc = S.pi * denr / 6.f;
    S.bvtr1 = 1.f + bvtr; S.bvtr2 = 2.5f + .5f * bvtr; S.bvtr3 = 3.f + bvtr; S.bvtr4 = 4.f + bvtr;
    S.g1pbr = rgmma(S.bvtr1); S.g3pbr = rgmma(S.bvtr3); S.g4pbr = rgmma(S.bvtr4); S.g5pbro2 = rgmma(S.bvtr2);
    S.pvtr = avtr * S.g4pbr / 6.f;
    S.eacrr = 1.0f;
    S.pacrr = S.pi * n0r * avtr * S.g3pbr * .25f * S.eacrr;
    S.precr1 = 2.f * S.pi * n0r * .78f;
    S.precr2 = 2.f * S.pi * n0r * .31f * powf(avtr, .5f) * S.g5pbro2;
    { const float r = dimax / dicon; S.xmmax = r * r; }
    { const float d2 = dimax * dimax, d4 = d2 * d2; S.roqimax = 2.08e22f * (d4 * d4); }
    S.bvts1 = 1.f + bvts; S.bvts2 = 2.5f + .5f * bvts; S.bvts3 = 3.f + bvts; S.bvts4 = 4.f + bvts;
    S.g1pbs = rgmma(S.bvts1); S.g3pbs = rgmma(S.bvts3); S.g4pbs = rgmma(S.bvts4); S.g5pbso2 = rgmma(S.bvts2);
    S.pvts = avts * S.g4pbs / 6.f;
    S.pacrs = S.pi * n0s * avts * S.g3pbs * .25f;
    S.precs1 = 4.f * n0s * .65f;
    S.precs2 = 4.f * n0s * .44f * powf(avts, .5f) * S.g5pbso2;
    S.pidn0r = S.pi * denr * n0r;
    S.pidn0s = S.pi * dens * n0s;
    S.rslopermax = 1.f / lamdarmax;
    S.rslopesmax = 1.f / lamdasmax;
    S.rsloperbmax = powf(S.rslopermax, bvtr);
    S.rslopesbmax = powf(S.rslopesmax, bvts);
    S.rsloper2max = S.rslopermax * S.rslopermax;
    S.rslopes2max = S.rslopesmax * S.rslopesmax;
    S.rsloper3max = S.rsloper2max * S.rslopermax;
    S.rslopes3max = S.rslopes2max * S.rslopesmax;
    { const float *p = (const float *)&S; for (int n = 0; n < (int)(sizeof S / sizeof(float)); ++n) out[n] = p[n]; }
}

/* slope_wsm3 :1008-1068 over n points (the reference's (its:ite, kts:kte) array section flattened: nothing in it couples points) */
static void slope_wsm3(int n, const float *qrs, const float *den, const float *denfac, const float *t, float *rslope, float *rslopeb,
                       float *rslope2, float *rslope3, float *vt)
{
    const float t0c = 273.15f;
    for (int c = 0; c < n; ++c) {
        float pvt;
        if (t[c] >= t0c) {
            pvt = S.pvtr;
            if (qrs[c] <= qcrmin) {
                rslope[c] = S.rslopermax; rslopeb[c] = S.rsloperbmax; rslope2[c] = S.rsloper2max; rslope3[c] = S.rsloper3max;
            } else {
                rslope[c] = 1.f / sqrtf(sqrtf(S.pidn0r / (qrs[c] * den[c])));                       /* lamdar */
                rslopeb[c] = x_exp(x_log(rslope[c]) * bvtr);
                rslope2[c] = rslope[c] * rslope[c];
                rslope3[c] = rslope2[c] * rslope[c];
            }
        } else {
            const float supcol = t0c - t[c];
            const float n0sfac = fmx(fmn(x_exp(alpha * supcol), n0smax / n0s), 1.f);
            pvt = S.pvts;
            if (qrs[c] <= qcrmin) {
                rslope[c] = S.rslopesmax; rslopeb[c] = S.rslopesbmax; rslope2[c] = S.rslopes2max; rslope3[c] = S.rslopes3max;
            } else {
                rslope[c] = 1.f / sqrtf(sqrtf(S.pidn0s * n0sfac / (qrs[c] * den[c])));              /* lamdas */
                rslopeb[c] = x_exp(x_log(rslope[c]) * bvts);
                rslope2[c] = rslope[c] * rslope[c];
                rslope3[c] = rslope2[c] * rslope[c];
            }
        }
        vt[c] = pvt * rslopeb[c] * denfac[c];
        if (qrs[c] <= 0.0f) vt[c] = 0.0f;
    }
}

#define MAXK 128
/* nislfv_rain_plm :1266-1504: the semi-Lagrangian fall of den*q through the columns of one slab.  Slab arrays X(i,k) sit at
 * X[(k-1)*im + (i-1)]; the locals below are 1-based like the reference's (element 0 unused). */
static void nislfv_rain_plm(int im, int km, const float *denl, const float *denfacl, const float *tkl, const float *dzl, const float *wwl,
                            float *rql, float *precip, float dt, int iter)
{
    float dz[MAXK + 2], ww[MAXK + 2], qq[MAXK + 2], wd[MAXK + 2], wa[MAXK + 2], was[MAXK + 2], den[MAXK + 2], denfac[MAXK + 2], tk[MAXK + 2];
    float wi[MAXK + 2], zi[MAXK + 2], za[MAXK + 2], qn[MAXK + 2], qr[MAXK + 2], tmp[MAXK + 2], tmp1[MAXK + 2], tmp2[MAXK + 2], tmp3[MAXK + 2];
    float dza[MAXK + 2], qa[MAXK + 2], qmi[MAXK + 2], qpi[MAXK + 2];
    for (int i = 0; i < im; ++i) precip[i] = 0.0f;
    for (int i = 0; i < im; ++i) {
        int k, n, kb, kt;
        float allold = 0.0f;
        for (k = 1; k <= km; ++k) {
            const size_t c = (size_t)(k - 1) * im + i;
            dz[k] = dzl[c]; qq[k] = rql[c]; ww[k] = wwl[c]; den[k] = denl[c]; denfac[k] = denfacl[c]; tk[k] = tkl[c];
        }
        for (k = 1; k <= km; ++k) allold = allold + qq[k];
        if (allold <= 0.0f) continue;                                       /* no precipitation in any layer */
        zi[1] = 0.0f;
        for (k = 1; k <= km; ++k) zi[k + 1] = zi[k] + dz[k];
        for (k = 1; k <= km; ++k) wd[k] = ww[k];                            /* departure wind */
        n = 1;
        for (;;) {                                                          /* label 100 */
            const float fa1 = 9.f / 16.f, fa2 = 1.f / 16.f, con1 = 0.05f;
            /* the 2nd-order wi of :1339-1343 is overwritten entirely by the 3rd-order one */
            wi[1] = ww[1];
            wi[2] = 0.5f * (ww[2] + ww[1]);
            for (k = 3; k <= km - 1; ++k) wi[k] = fa1 * (ww[k] + ww[k - 1]) - fa2 * (ww[k + 1] + ww[k - 2]);
            wi[km] = 0.5f * (ww[km] + ww[km - 1]);
            wi[km + 1] = ww[km];
            for (k = 2; k <= km; ++k) if (ww[k] == 0.0f) wi[k] = ww[k - 1];  /* top of the rain group */
            for (k = km; k >= 1; --k) {                                     /* diffusivity of wi */
                const float decfl = (wi[k + 1] - wi[k]) * dt / dz[k];
                if (decfl > con1) wi[k] = wi[k + 1] - con1 * dz[k] / dt;
            }
            for (k = 1; k <= km + 1; ++k) za[k] = zi[k] - wi[k] * dt;       /* arrival points */
            for (k = 1; k <= km; ++k) dza[k] = za[k + 1] - za[k];
            dza[km + 1] = zi[km + 1] - za[km + 1];
            for (k = 1; k <= km; ++k) { qa[k] = qq[k] * dz[k] / dza[k]; qr[k] = qa[k] / den[k]; }
            qa[km + 1] = 0.0f;
            if (n > iter) break;
            /* arrival terminal velocity, then the mean of departure and arrival winds */
            slope_wsm3(km, qr + 1, den + 1, denfac + 1, tk + 1, tmp + 1, tmp1 + 1, tmp2 + 1, tmp3 + 1, wa + 1);
            if (n >= 2) for (k = 1; k <= km; ++k) wa[k] = 0.5f * (wa[k] + was[k]);
            for (k = 1; k <= km; ++k) ww[k] = 0.5f * (wd[k] + wa[k]);
            for (k = 1; k <= km; ++k) was[k] = wa[k];
            n = n + 1;
        }
        /* monotone piecewise-linear values at the arrival cell interfaces */
        for (k = 2; k <= km; ++k) {
            const float dip = (qa[k + 1] - qa[k]) / (dza[k + 1] + dza[k]);
            const float dim = (qa[k] - qa[k - 1]) / (dza[k - 1] + dza[k]);
            if (dip * dim <= 0.0f) {
                qmi[k] = qa[k]; qpi[k] = qa[k];
            } else {
                qpi[k] = qa[k] + 0.5f * (dip + dim) * dza[k];
                qmi[k] = 2.0f * qa[k] - qpi[k];
                if (qpi[k] < 0.0f || qmi[k] < 0.0f) { qpi[k] = qa[k]; qmi[k] = qa[k]; }
            }
        }
        qpi[1] = qa[1]; qmi[1] = qa[1]; qmi[km + 1] = qa[km + 1]; qpi[km + 1] = qa[km + 1];
        /* interpolation to the regular grid */
        for (k = 1; k <= km; ++k) qn[k] = 0.0f;
        kb = 1; kt = 1;
        for (k = 1; k <= km; ++k) {
            int kk, m;
            kb = kb - 1 > 1 ? kb - 1 : 1;
            kt = kt - 1 > 1 ? kt - 1 : 1;
            if (zi[k] >= za[km + 1]) break;
            for (kk = kb; kk <= km; ++kk) if (zi[k] <= za[kk + 1]) { kb = kk; break; }
            for (kk = kt; kk <= km; ++kk) if (zi[k + 1] <= za[kk]) { kt = kk; break; }
            kt = kt - 1;
            if (kt == kb) {
                const float tl = (zi[k] - za[kb]) / dza[kb], th = (zi[k + 1] - za[kb]) / dza[kb];
                const float tl2 = tl * tl, th2 = th * th;
                const float qqd = 0.5f * (qpi[kb] - qmi[kb]);
                const float qqh = qqd * th2 + qmi[kb] * th, qql = qqd * tl2 + qmi[kb] * tl;
                qn[k] = (qqh - qql) / (th - tl);
            } else if (kt > kb) {
                float tl = (zi[k] - za[kb]) / dza[kb];
                float tl2 = tl * tl;
                float qqd = 0.5f * (qpi[kb] - qmi[kb]);
                const float qql = qqd * tl2 + qmi[kb] * tl;
                const float dql = qa[kb] - qql;
                float zsum = (1.f - tl) * dza[kb];
                float qsum = dql * dza[kb];
                float th, th2, dqh;
                if (kt - kb > 1) for (m = kb + 1; m <= kt - 1; ++m) { zsum = zsum + dza[m]; qsum = qsum + qa[m] * dza[m]; }
                th = (zi[k + 1] - za[kt]) / dza[kt];
                th2 = th * th;
                qqd = 0.5f * (qpi[kt] - qmi[kt]);
                dqh = qqd * th2 + qmi[kt] * th;
                zsum = zsum + th * dza[kt];
                qsum = qsum + dqh * dza[kt];
                qn[k] = qsum / zsum;
            }
        }
        /* rain out */
        for (k = 1; k <= km; ++k) {
            if (za[k] < 0.0f && za[k + 1] < 0.0f) { precip[i] = precip[i] + qa[k] * dza[k]; continue; }
            else if (za[k] < 0.0f && za[k + 1] >= 0.0f) { precip[i] = precip[i] + qa[k] * (0.0f - za[k]); break; }
            break;
        }
        for (k = 1; k <= km; ++k) rql[(size_t)(k - 1) * im + i] = qn[k];
    }
}

/* what mp_driver.f90:554-585 passes, in its order */
typedef struct { float delt, g, cpd, cpv, rd, rv, t0c, ep1, ep2, qmin, xls, xlv0, xlf0, den0, denr, cliq, cice, psat; } w3_args;

/* wsm32D :218-903 on one slab.  X(i,k), i = its..ite, k = kts..kte (kts = 1, as the reference's fall(i,1) assumes), sits at
 * X[(k-1)*im + (i-its)]; the macro A(X,k) below is X(i,k) for the loop's i and a 1-based k.  Returns 1 when a column is warm
 * to its top level with rising air there (the reference then reads w, qrs, qci one level above kte). */
static int wsm32D(const w3_args *a, int im, int km, float *t, float *q, float *qci, float *qrs, const float *w, const float *den,
                  const float *p, const float *delz, float *rain, float *rainncv, float *snow, float *snowncv, float *sr, float *work)
{
    const size_t n = (size_t)im * km;
    const float delt = a->delt, cpd = a->cpd, cpv = a->cpv, rv = a->rv, t0c = a->t0c, ep2 = a->ep2, qmin = a->qmin, xls = a->xls,
                xlv0 = a->xlv0, xlf0 = a->xlf0, den0 = a->den0, denr = a->denr, cliq = a->cliq, cice = a->cice, psat = a->psat;
    float *rh = work, *qs = rh + n, *denfac = qs + n, *rslope = denfac + n, *rslope2 = rslope + n, *rslope3 = rslope2 + n,
          *qrs_tmp = rslope3 + n, *den_tmp = qrs_tmp + n, *delz_tmp = den_tmp + n, *rslopeb = delz_tmp + n, *pgen = rslopeb + n,
          *pisd = pgen + n, *paut = pisd + n, *pacr = paut + n, *pres = pacr + n, *pcon = pres + n, *fall = pcon + n, *xl = fall + n,
          *cpm = xl + n, *work1 = cpm + n, *work2 = work1 + n, *xni = work2 + n, *qs0 = xni + n, *denqci = qs0 + n, *denqrs = denqci + n,
          *n0sfac = denqrs + n, *work1c = n0sfac + n, *fallc = work1c + n, *delqrs = fallc + n, *delqi = delqrs + im,
          *tstepsnow = delqi + im;
    int *mstep = (int *)(tstepsnow + im), *kwork1 = mstep + im, *kwork2 = kwork1 + im;
    int loops, loop, i, k, bad = 0;
    float dtcld, cvap, hvap, hsub, ttp, dldt, xa, xb, dldti, xai, xbi;
#define A(X, k) X[(size_t)((k) - 1) * im + i]
    /* the statement functions of :336-346, operand order as written */
#define cpmcal(x) (cpd * (1.f - fmx(x, qmin)) + fmx(x, qmin) * cpv)
#define xlcal(x) (xlv0 - S.xlv1 * ((x) - t0c))
#define diffus(x, y) (8.794e-5f * x_exp(x_log(x) * 1.81f) / (y))
#define viscos(x, y) (1.496e-6f * ((x) * sqrtf(x)) / ((x) + 120.f) / (y))
#define xka(x, y) (1.414e3f * viscos(x, y) * (y))
#define diffac(a_, b_, c_, d_, e_) ((d_) * (a_) * (a_) / (xka(c_, d_) * rv * (c_) * (c_)) + 1.f / ((e_) * diffus(c_, b_)))
#define venfac(a_, b_, c_) (x_exp(x_log(viscos(b_, c_) / diffus(b_, a_)) * .3333333f) / sqrtf(viscos(b_, c_)) * sqrtf(sqrtf(den0 / (c_))))
#define conden(a_, b_, c_, d_, e_) ((fmx(b_, qmin) - (c_)) / (1.f + (d_) * (d_) / (rv * (e_)) * (c_) / ((a_) * (a_))))
    for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) { A(qci, k) = fmx(A(qci, k), 0.0f); A(qrs, k) = fmx(A(qrs, k), 0.0f); }   /* padding for small values */
    for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) { A(cpm, k) = cpmcal(A(q, k)); A(xl, k) = xlcal(A(t, k)); }            /* latent heat, heat capacity */
    for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) { A(delz_tmp, k) = A(delz, k); A(den_tmp, k) = A(den, k); }
    for (i = 0; i < im; ++i) { rainncv[i] = 0.f; snowncv[i] = 0.f; sr[i] = 0.f; tstepsnow[i] = 0.f; }
    /* minor time steps */
    loops = (int)lroundf(delt / dtcldcr); if (loops < 1) loops = 1;
    dtcld = delt / (float)loops;
    if (delt <= dtcldcr) dtcld = delt;
    for (loop = 1; loop <= loops; ++loop) {
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            float tv = 1.0f / A(den, k);
            tv = tv * den0;
            A(denfac, k) = sqrtf(tv);
        }
        /* inline fpvs */
        cvap = cpv; hvap = xlv0; hsub = xls; ttp = t0c + 0.01f;
        dldt = cvap - cliq; xa = -dldt / rv; xb = xa + hvap / (rv * ttp);
        dldti = cvap - cice; xai = -dldti / rv; xbi = xai + hsub / (rv * ttp);
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            const float tr = ttp / A(t, k);
            if (A(t, k) < ttp) A(qs, k) = psat * x_exp(x_log(tr) * xai) * x_exp(xbi * (1.f - tr));
            else A(qs, k) = psat * x_exp(x_log(tr) * xa) * x_exp(xb * (1.f - tr));
            A(qs0, k) = psat * x_exp(x_log(tr) * xa) * x_exp(xb * (1.f - tr));
            A(qs0, k) = (A(qs0, k) - A(qs, k)) / A(qs, k);
            A(qs, k) = fmn(A(qs, k), 0.99f * A(p, k));
            A(qs, k) = ep2 * A(qs, k) / (A(p, k) - A(qs, k));
            A(qs, k) = fmx(A(qs, k), qmin);
            A(rh, k) = fmx(A(q, k) / A(qs, k), qmin);
        }
        /* initialise the production rates */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            A(pres, k) = 0.f; A(paut, k) = 0.f; A(pacr, k) = 0.f; A(pgen, k) = 0.f; A(pisd, k) = 0.f; A(pcon, k) = 0.f;
            A(fall, k) = 0.f; A(fallc, k) = 0.f; A(xni, k) = 1.e3f;
        }
        /* ice crystal number concentration */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i)
            A(xni, k) = fmn(fmx(5.38e7f * x_exp(x_log(A(den, k) * fmx(A(qci, k), qmin)) * 0.75f), 1.e3f), 1.e6f);
        /* fall of rain / snow */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) A(qrs_tmp, k) = A(qrs, k);
        slope_wsm3((int)n, qrs_tmp, den_tmp, denfac, t, rslope, rslopeb, rslope2, rslope3, work1);
        for (k = km; k >= 1; --k) for (i = 0; i < im; ++i) A(denqrs, k) = A(den, k) * A(qrs, k);
        nislfv_rain_plm(im, km, den_tmp, denfac, t, delz_tmp, work1, denqrs, delqrs, dtcld, 1);
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            A(qrs, k) = fmx(A(denqrs, k) / A(den, k), 0.f);
            A(fall, k) = A(denqrs, k) * A(work1, k) / A(delz, k);
        }
        for (i = 0; i < im; ++i) A(fall, 1) = delqrs[i] / A(delz, 1) / dtcld;
        /* fall of cloud ice */
        for (k = km; k >= 1; --k) for (i = 0; i < im; ++i) {
            if (A(t, k) < t0c && A(qci, k) > 0.f) {
                const float xmi = A(den, k) * A(qci, k) / A(xni, k);
                const float diameter = fmx(dicon * sqrtf(xmi), 1.e-25f);
                A(work1c, k) = 1.49e4f * x_exp(x_log(diameter) * 1.31f);
            } else A(work1c, k) = 0.f;
        }
        for (k = km; k >= 1; --k) for (i = 0; i < im; ++i) A(denqci, k) = A(den, k) * A(qci, k);
        nislfv_rain_plm(im, km, den_tmp, denfac, t, delz_tmp, work1c, denqci, delqi, dtcld, 0);
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) A(qci, k) = fmx(A(denqci, k) / A(den, k), 0.f);
        for (i = 0; i < im; ++i) A(fallc, 1) = delqi[i] / A(delz, 1) / dtcld;
        /* melting / freezing at the highest warm level */
        for (i = 0; i < im; ++i) mstep[i] = 0;
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) if (A(t, k) >= t0c) mstep[i] = k;
        for (i = 0; i < im; ++i) {
            kwork2[i] = mstep[i]; kwork1[i] = mstep[i];
            if (mstep[i] != 0) { if (A(w, mstep[i]) > 0.f) kwork1[i] = mstep[i] + 1; }
        }
        for (i = 0; i < im; ++i) {
            const int kk = kwork2[i];
            k = kwork1[i];
            if (k > km) { bad = 1; continue; }
            if (k * kk >= 1) {
                const float qrsci = A(qrs, k) + A(qci, k);
                if (qrsci > 0.f || A(fall, kk) > 0.f) {
                    const float frzmlt = fmn(fmx(-A(w, k) * qrsci / A(delz, k), -qrsci / dtcld), qrsci / dtcld);
                    const float snomlt = fmn(fmx(A(fall, kk) / A(den, kk), -A(qrs, k) / dtcld), A(qrs, k) / dtcld);
                    if (k == kk) A(t, k) = A(t, k) - xlf0 / A(cpm, k) * (frzmlt + snomlt) * dtcld;
                    else {
                        A(t, k) = A(t, k) - xlf0 / A(cpm, k) * frzmlt * dtcld;
                        A(t, kk) = A(t, kk) - xlf0 / A(cpm, kk) * snomlt * dtcld;
                    }
                }
            }
        }
        /* what reaches the surface */
        for (i = 0; i < im; ++i) {
            float fallsum = A(fall, 1), fallsum_qsi = 0.f;
            if ((t0c - A(t, 1)) > 0) { fallsum = fallsum + A(fallc, 1); fallsum_qsi = A(fall, 1) + A(fallc, 1); }
            if (fallsum > 0.f) {
                rainncv[i] = fallsum * A(delz, 1) / denr * dtcld * 1000.f + rainncv[i];
                rain[i] = fallsum * A(delz, 1) / denr * dtcld * 1000.f + rain[i];
            }
            if (fallsum_qsi > 0.f) {
                tstepsnow[i] = fallsum_qsi * A(delz, 1) / denr * dtcld * 1000.f + tstepsnow[i];
                snowncv[i] = fallsum_qsi * A(delz, 1) / denr * dtcld * 1000.f + snowncv[i];
                snow[i] = fallsum_qsi * A(delz, 1) / denr * dtcld * 1000.f + snow[i];
            }
            if (fallsum > 0.f) sr[i] = snowncv[i] / (rainncv[i] + 1.e-12f);
        }
        /* slopes again, with the fallen qrs */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) A(qrs_tmp, k) = A(qrs, k);
        slope_wsm3((int)n, qrs_tmp, den_tmp, denfac, t, rslope, rslopeb, rslope2, rslope3, work1);
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            if (A(t, k) >= t0c) A(work1, k) = diffac(A(xl, k), A(p, k), A(t, k), A(den, k), A(qs, k));
            else A(work1, k) = diffac(xls, A(p, k), A(t, k), A(den, k), A(qs, k));
            A(work2, k) = venfac(A(p, k), A(t, k), A(den, k));
        }
        /* the production rates: warm rain above 0 C, cold rain below */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            const float supsat = fmx(A(q, k), qmin) - A(qs, k);
            const float satdt = supsat / dtcld;
            if (A(t, k) >= t0c) {
                if (A(qci, k) > S.qc0) {                                                                   /* praut */
                    A(paut, k) = S.qck1 * x_exp(x_log(A(qci, k)) * (7.f / 3.f));
                    A(paut, k) = fmn(A(paut, k), A(qci, k) / dtcld);
                }
                if (A(qrs, k) > qcrmin && A(qci, k) > qmin)                                                /* pracw */
                    A(pacr, k) = fmn(S.pacrr * A(rslope3, k) * A(rslopeb, k) * A(qci, k) * A(denfac, k), A(qci, k) / dtcld);
                if (A(qrs, k) > 0.f) {                                                                     /* prevp */
                    const float coeres = A(rslope2, k) * sqrtf(A(rslope, k) * A(rslopeb, k));
                    A(pres, k) = (A(rh, k) - 1.f) * (S.precr1 * A(rslope2, k) + S.precr2 * A(work2, k) * coeres) / A(work1, k);
                    if (A(pres, k) < 0.f) {
                        A(pres, k) = fmx(A(pres, k), -A(qrs, k) / dtcld);
                        A(pres, k) = fmx(A(pres, k), satdt / 2.f);
                    } else A(pres, k) = fmn(A(pres, k), satdt / 2.f);
                }
            } else {
                const float supcol = t0c - A(t, k);
                int ifsat = 0;
                float eacrs;
                A(n0sfac, k) = fmx(fmn(x_exp(alpha * supcol), n0smax / n0s), 1.f);
                A(xni, k) = fmn(fmx(5.38e7f * x_exp(x_log(A(den, k) * fmx(A(qci, k), qmin)) * 0.75f), 1.e3f), 1.e6f);
                eacrs = x_exp(0.07f * (-supcol));
                if (A(qrs, k) > qcrmin && A(qci, k) > qmin) {                                              /* psaci */
                    const float xmi = A(den, k) * A(qci, k) / A(xni, k);
                    const float diameter = fmn(dicon * sqrtf(xmi), dimax);
                    const float vt2i = 1.49e4f * x_pow(diameter, 1.31f);
                    const float vt2s = S.pvts * A(rslopeb, k) * A(denfac, k);
                    const float acrfac = 2.f * A(rslope3, k) + 2.f * diameter * A(rslope2, k) + diameter * diameter * A(rslope, k);
                    A(pacr, k) = fmn(S.pi * A(qci, k) * eacrs * n0s * A(n0sfac, k) * fabsf(vt2s - vt2i) * acrfac / 4.f, A(qci, k) / dtcld);
                }
                if (A(qci, k) > 0.f) {                                                                     /* pisd */
                    const float xmi = A(den, k) * A(qci, k) / A(xni, k);
                    const float diameter = dicon * sqrtf(xmi);
                    A(pisd, k) = 4.f * diameter * A(xni, k) * (A(rh, k) - 1.f) / A(work1, k);
                    if (A(pisd, k) < 0.f) {
                        A(pisd, k) = fmx(A(pisd, k), satdt / 2.f);
                        A(pisd, k) = fmx(A(pisd, k), -A(qci, k) / dtcld);
                    } else A(pisd, k) = fmn(A(pisd, k), satdt / 2.f);
                    if (fabsf(A(pisd, k)) >= fabsf(satdt)) ifsat = 1;
                }
                if (A(qrs, k) > 0.f && ifsat != 1) {                                                       /* psdep */
                    const float coeres = A(rslope2, k) * sqrtf(A(rslope, k) * A(rslopeb, k));
                    float supice;
                    A(pres, k) = (A(rh, k) - 1.f) * A(n0sfac, k) * (S.precs1 * A(rslope2, k) + S.precs2 * A(work2, k) * coeres) / A(work1, k);
                    supice = satdt - A(pisd, k);
                    if (A(pres, k) < 0.f) {
                        A(pres, k) = fmx(A(pres, k), -A(qrs, k) / dtcld);
                        A(pres, k) = fmx(fmx(A(pres, k), satdt / 2.f), supice);
                    } else A(pres, k) = fmn(fmn(A(pres, k), satdt / 2.f), supice);
                    if (fabsf(A(pisd, k) + A(pres, k)) >= fabsf(satdt)) ifsat = 1;
                }
                if (supsat > 0 && ifsat != 1) {                                                            /* pigen */
                    const float supice = satdt - A(pisd, k) - A(pres, k);
                    const float xni0 = 1.e3f * x_exp(0.1f * supcol);
                    const float roqi0 = 4.92e-11f * x_exp(x_log(xni0) * 1.33f);
                    A(pgen, k) = fmx(0.f, (roqi0 / A(den, k) - fmx(A(qci, k), 0.f)) / dtcld);
                    A(pgen, k) = fmn(fmn(A(pgen, k), satdt), supice);
                }
                if (A(qci, k) > 0.f) {                                                                     /* psaut */
                    const float qimax = S.roqimax / A(den, k);
                    A(paut, k) = fmx(0.f, (A(qci, k) - qimax) / dtcld);
                }
            }
        }
        /* feasibility of the rates, then the update */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            const float qciik = fmx(qmin, A(qci, k));
            const float delqci = (A(paut, k) + A(pacr, k) - A(pgen, k) - A(pisd, k)) * dtcld;
            float qik, delq;
            if (delqci >= qciik) {
                const float facqci = qciik / delqci;
                A(paut, k) = A(paut, k) * facqci; A(pacr, k) = A(pacr, k) * facqci; A(pgen, k) = A(pgen, k) * facqci; A(pisd, k) = A(pisd, k) * facqci;
            }
            qik = fmx(qmin, A(q, k));
            delq = (A(pres, k) + A(pgen, k) + A(pisd, k)) * dtcld;
            if (delq >= qik) {
                const float facq = qik / delq;
                A(pres, k) = A(pres, k) * facq; A(pgen, k) = A(pgen, k) * facq; A(pisd, k) = A(pisd, k) * facq;
            }
            A(work2, k) = -A(pres, k) - A(pgen, k) - A(pisd, k);
            A(q, k) = A(q, k) + A(work2, k) * dtcld;
            A(qci, k) = fmx(A(qci, k) - (A(paut, k) + A(pacr, k) - A(pgen, k) - A(pisd, k)) * dtcld, 0.f);
            A(qrs, k) = fmx(A(qrs, k) + (A(paut, k) + A(pacr, k) + A(pres, k)) * dtcld, 0.f);
            if (A(t, k) < t0c) A(t, k) = A(t, k) - xls * A(work2, k) / A(cpm, k) * dtcld;
            else A(t, k) = A(t, k) - A(xl, k) * A(work2, k) / A(cpm, k) * dtcld;
        }
        /* saturation over water at the new temperature */
        cvap = cpv; hvap = xlv0; hsub = xls; ttp = t0c + 0.01f;
        dldt = cvap - cliq; xa = -dldt / rv; xb = xa + hvap / (rv * ttp);
        dldti = cvap - cice; xai = -dldti / rv; xbi = xai + hsub / (rv * ttp);
        (void)xbi;
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            const float tr = ttp / A(t, k);
            A(qs, k) = psat * x_exp(x_log(tr) * xa) * x_exp(xb * (1.f - tr));
            A(qs, k) = fmn(A(qs, k), 0.99f * A(p, k));
            A(qs, k) = ep2 * A(qs, k) / (A(p, k) - A(qs, k));
            A(qs, k) = fmx(A(qs, k), qmin);
            A(denfac, k) = sqrtf(den0 / A(den, k));
        }
        /* pcond: condensation / evaporation of cloud water */
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            A(work1, k) = conden(A(t, k), A(q, k), A(qs, k), A(xl, k), A(cpm, k));
            A(work2, k) = A(qci, k) + A(work1, k);
            A(pcon, k) = fmn(fmx(A(work1, k), 0.f), fmx(A(q, k), 0.f)) / dtcld;
            if (A(qci, k) > 0.f && A(work1, k) < 0 && A(t, k) > t0c) A(pcon, k) = fmx(A(work1, k), -A(qci, k)) / dtcld;
            A(q, k) = A(q, k) - A(pcon, k) * dtcld;
            A(qci, k) = fmx(A(qci, k) + A(pcon, k) * dtcld, 0.f);
            A(t, k) = A(t, k) + A(pcon, k) * A(xl, k) / A(cpm, k) * dtcld;
        }
        for (k = 1; k <= km; ++k) for (i = 0; i < im; ++i) {
            if (A(qci, k) <= qmin) A(qci, k) = 0.0f;
            if (A(qrs, k) <= qcrmin) A(qrs, k) = 0.0f;
        }
    }
#undef A
    return bad;
}
#define W3_SLABS 28     /* the (i,k) work arrays wsm32D carves out of its buffer */

/* wsm3 (:74-216): t = th*pii, wsm32D on the (its:ite, kts:kte) slab of each row j, th = t/pii.  Arrays X(i,k,j) -> i + nx*(k + nz*j),
 * 1-based inclusive tile bounds.  Returns 0, 1 for a tile this file cannot take, 2 for the out-of-range read described at wsm32D. */
int orc_wsm3(int nx, int nz, int ny, float *th, float *q, float *qci, float *qrs, const float *w, const float *den, const float *pii,
             const float *p, const float *delz, const float *args18, float *rain, float *rainncv, float *snow, float *snowncv, float *sr,
             int its, int ite, int jts, int jte, int kts, int kte)
{
    w3_args a;
    const int im = ite - its + 1, km = kte - kts + 1;
    int rc = 0;
    if (km > MAXK || km < 3 || kts != 1 || im < 1) return 1;
    { float *f = (float *)&a; for (int n = 0; n < 18; ++n) f[n] = args18[n]; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = jts - 1; j <= jte - 1; ++j) {
        const size_t n = (size_t)im * km;
        /* the slab sections q(its:ite, kts:kte, j) ... gathered contiguous, + wsm32D's work arrays */
        float *buf = (float *)malloc(sizeof(float) * ((9 + W3_SLABS) * n + 6 * (size_t)im));
        float *t = buf, *sq = t + n, *sqci = sq + n, *sqrs = sqci + n, *sw = sqrs + n, *sden = sw + n, *sp = sden + n, *sdz = sp + n, *work = sdz + n;
        const size_t o = (size_t)(its - 1) + (size_t)nx * j;
        for (int k = 0; k < km; ++k) for (int i = 0; i < im; ++i) {
            const size_t c = (size_t)(i + its - 1) + (size_t)nx * ((size_t)k + (size_t)nz * j), s = (size_t)k * im + i;
            t[s] = th[c] * pii[c]; sq[s] = q[c]; sqci[s] = qci[c]; sqrs[s] = qrs[c]; sw[s] = w[c]; sden[s] = den[c]; sp[s] = p[c]; sdz[s] = delz[c];
        }
        if (wsm32D(&a, im, km, t, sq, sqci, sqrs, sw, sden, sp, sdz, rain + o, rainncv + o, snow + o, snowncv + o, sr + o, work)) {
#pragma omp atomic write
            rc = 2;
        }
        for (int k = 0; k < km; ++k) for (int i = 0; i < im; ++i) {
            const size_t c = (size_t)(i + its - 1) + (size_t)nx * ((size_t)k + (size_t)nz * j), s = (size_t)k * im + i;
            th[c] = t[s] / pii[c]; q[c] = sq[s]; qci[c] = sqci[s]; qrs[c] = sqrs[s];
        }
        free(buf);
    }
    return rc;
}
