// icar_amd/csrc/glibc_dbl64.h -- exp / log / pow in DOUBLE PRECISION exactly as the compiled reference evaluates them.
// Origin and licence: restated from the GNU C Library 2.35 (sysdeps/ieee754/dbl-64/e_exp.c, e_log.c, e_pow.c and their *_data.c),
// Copyright (C) Free Software Foundation, Inc. / Arm Ltd. ("optimized routines"), distributed under the GNU Lesser General Public
// License, version 2.1 or later (the Arm originals also under the MIT licence).  The algorithms and table constants are theirs;
// this file is a derived work under the same terms.
//
// The Thompson scheme's DOUBLE PRECISION sites (mp_thompson.f90: N0_r, N0_g, lam_exp, the collection / evaporation / melting
// integrals, the bin indices) call the C library's pow / log / exp.  Until round 4 the device evaluated them with its own
// FP64 polynomials (fp64_math.h: < 1 ulp of the double, but not the library's bits); the results pass through float roundings and
// threshold tests, and ~1e-7 of the cells of a 512 x 512 x 40 step came out one float ulp away from the reference (found by the
// whole-loop test of round 4; profiles/r04_steps.md).  This header restates glibc 2.35's functions (not vendored in
// /root/reference): sysdeps/ieee754/dbl-64/e_exp.c, e_log.c, e_pow.c -- the "ARM optimized routines" algorithms, one table
// look-up + a polynomial, with the data of e_exp_data.c / e_log_data.c / e_pow_log_data.c (glibc_dbl64_tables.h).
//
// x86-64 glibc selects the FMA builds (__exp_fma, __log_fma, __pow_fma) on any CPU with AVX2 + FMA; the compiler contracted
// a*b+c there, and WHICH products were fused decides the last bit.  The sequences below were read off the image's libm.so.6
// (objdump -d) operation by operation; every fma() is a vfmadd of that code and every separate * and + is a vmulsd / vaddsd.
// Checked value by value against the host libm on the CPU (tests/glibc_dbl64_check.cpp, tests/test_glibc_dbl64_host.py) and
// on the MI355X (tests/test_gpu_glibc_math.py).
//
// Only the main paths matter to the microphysics (positive normal bases, moderate exponents); the special cases (zero,
// negative, subnormal, inf, NaN, overflow, underflow) follow the published source's case analysis and are covered by the same
// checks.  errno / exception flags are not modelled.
#pragma once
#include <stdint.h>
#ifndef GD_FN
#define GD_FN __device__ __forceinline__
#define GD_TABLE __device__ const
#endif
#ifndef GD_UNLIKELY
#define GD_UNLIKELY(c) __builtin_expect(!!(c), 0)
#endif
#include "glibc_dbl64_tables.h"

// Tables: GD_LDS_TABLES defined before the include -> the tables of exp and of pow's logarithm (5 KB) live in LDS (`gd_lds`; every kernel of that
// translation unit that calls these functions runs gd_lds_init() and a barrier first); the 35 scalar coefficients stay in constant
// global memory (scalar loads).  Otherwise the tables are read from constant global memory too.
#ifdef GD_LDS_TABLES
struct GdLds { uint64_t exp_tab[256]; double powlog_tab[384]; };      // (log's own table -- two call sites -- stays in global memory)
__shared__ GdLds gd_lds;
// every thread of the block calls this once at the top of the kernel; the CALLER's barrier must follow
__device__ __forceinline__ void gd_lds_init(int tid, int nthreads)
{
    const double2 *se = (const double2 *)gd_data.exp_tab, *sp = (const double2 *)gd_data.powlog_tab;
    double2 *de = (double2 *)gd_lds.exp_tab, *dp = (double2 *)gd_lds.powlog_tab;
    for (int t = tid; t < 128; t += nthreads) de[t] = se[t];
    for (int t = tid; t < 192; t += nthreads) dp[t] = sp[t];
}
#define GD_T gd_lds
#define GD_TLOG gd_data
#else
#define GD_T gd_data
#define GD_TLOG gd_data
#endif

GD_FN uint64_t gd_asuint64(double f) { uint64_t u; __builtin_memcpy(&u, &f, 8); return u; }
GD_FN double gd_asdouble(uint64_t u) { double f; __builtin_memcpy(&f, &u, 8); return f; }
GD_FN uint32_t gd_top12(double x) { return (uint32_t)(gd_asuint64(x) >> 52); }
#define GD_INF 0x7ff0000000000000ull
#define GD_ONE 0x3ff0000000000000ull

// ---- e_exp.c: specialcase() -- the exponent of scale may over- / underflow (|x| in [512, 1024)) ----
GD_FN double gd_exp_special(double tmp, uint64_t sbits, uint64_t ki)
{
    if ((ki & 0x80000000ull) == 0) {                        // k > 0
        sbits -= 1009ull << 52;
        const double scale = gd_asdouble(sbits);
        return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                 // k < 0: care in the subnormal range
    const double scale = gd_asdouble(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (__builtin_fabs(y) < 1.0) {
        double one = 1.0;
        if (y < 0.0) one = -1.0;
        double lo = scale - y + st;
        const double hi = one + y;
        lo = one - hi + y + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = gd_asdouble(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
}

// exp(x + xtail) * (-1)^(sign_bias != 0) for |x| >= 2^-54: the shared core of __exp (xtail = 0: `r += xtail` is absent there)
// and of pow's exp_inline.  abstop = 0 marks |x| in [512, 1024).
template <bool TAIL>
GD_FN double gd_exp_core(double x, double xtail, uint64_t sign_bias, uint32_t abstop)
{
    const double *E = gd_data.exp_hdr;
    const double kdr = __builtin_fma(x, E[0], E[1]);        // InvLn2N * x + Shift, fused
    const uint64_t ki = gd_asuint64(kdr);
    const double kd = kdr - E[1];
    double r = __builtin_fma(kd, E[2], x);                  // x + kd * NegLn2hiN
    r = __builtin_fma(kd, E[3], r);                         //   + kd * NegLn2loN
    if (TAIL) r = xtail + r;
    const uint64_t idx = 2 * (ki & 127);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = gd_asdouble(GD_T.exp_tab[idx]);
    const uint64_t sbits = GD_T.exp_tab[idx + 1] + top;
    const double p23 = __builtin_fma(r, E[5], E[4]);        // C2 + r C3
    const double rt = r + tail;
    const double r2 = r * r;
    const double p45 = __builtin_fma(r, E[7], E[6]);        // C4 + r C5
    const double lowp = __builtin_fma(p23, r2, rt);         // tail + r + r2 (C2 + r C3)
    const double tmp = __builtin_fma(r2 * r2, p45, lowp);
    if (GD_UNLIKELY(abstop == 0)) return gd_exp_special(tmp, sbits, ki);
    const double scale = gd_asdouble(sbits);
    return __builtin_fma(scale, tmp, scale);
}

GD_FN double gd_exp(double x)
{
    uint32_t abstop = gd_top12(x) & 0x7ff;
    if (GD_UNLIKELY(abstop - 0x3c9 >= 0x3f)) {              // |x| < 2^-54 or |x| >= 512 or NaN
        if (abstop - 0x3c9 >= 0x80000000u) return 1.0 + x;
        if (abstop >= 0x409) {                              // |x| >= 1024, inf, NaN
            if (gd_asuint64(x) == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ff) return 1.0 + x;
            return (gd_asuint64(x) >> 63) ? 0.0 : gd_asdouble(GD_INF);
        }
        abstop = 0;
    }
    return gd_exp_core<false>(x, 0.0, 0, abstop);
}

GD_FN double gd_log(double x)
{
    const double *L = gd_data.log_hdr, *A = L + 2, *B = L + 7;
    uint64_t ix = gd_asuint64(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (GD_UNLIKELY(ix - 0x3fee000000000000ull < 0x3090000000000ull)) {          // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == GD_ONE) return 0.0;
        const double r = x - 1.0;
        const double q1 = __builtin_fma(r, B[2], B[1]), q4 = __builtin_fma(r, B[5], B[4]), q7 = __builtin_fma(r, B[8], B[7]);
        const double r2 = r * r;
        const double p1 = __builtin_fma(r2, B[3], q1), p4 = __builtin_fma(r2, B[6], q4);
        const double r3 = r * r2;
        double p7 = __builtin_fma(r2, B[9], q7);
        p7 = __builtin_fma(r3, B[10], p7);
        double pol = __builtin_fma(p7, r3, p4);
        pol = __builtin_fma(pol, r3, p1);                    // B1 + r B2 + r2 B3 + r3 (B4 + ... + r3 (B7 + ... + r3 B10))
        const double rw = __builtin_fma(r, 0x1p27, r);       // r + w, w = r 2^27
        const double rhi = __builtin_fma(-0x1p27, r, rw);    // r + w - w
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = __builtin_fma(rhi2, B[0], r);      // r + rhi rhi B0
        const double rmh = r - hi;
        const double rs = r + rhi;
        double lo = __builtin_fma(rhi2, B[0], rmh);          // r - hi + w
        lo = __builtin_fma(B[0] * rlo, rs, lo);
        const double y = __builtin_fma(pol, r3, lo);
        return y + hi;
    }
    if (GD_UNLIKELY(top - 0x0010 >= 0x7ff0 - 0x0010)) {      // x < 2^-1022, inf, NaN
        if (ix * 2 == 0) return -gd_asdouble(GD_INF);
        if (ix == GD_INF) return x;
        if ((top & 0x8000) || (top & 0x7ff0) == 0x7ff0) return (x - x) / 0.0;      // negative: NaN; NaN stays
        ix = gd_asuint64(x * 0x1p52);
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double invc = GD_TLOG.log_tab[2 * i], logc = GD_TLOG.log_tab[2 * i + 1];
    const double z = gd_asdouble(iz);
    const double kd = (double)k;
    const double r = __builtin_fma(z, invc, -1.0);
    const double w = __builtin_fma(kd, L[0], logc);
    const double q12 = __builtin_fma(r, A[2], A[1]);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, L[1], lo);
    const double r3 = r * r2;
    const double q34 = __builtin_fma(r, A[4], A[3]);
    lo = __builtin_fma(r2, A[0], lo);
    const double q = __builtin_fma(q34, r2, q12);
    const double y = __builtin_fma(r3, q, lo);
    return y + hi;
}

// ---- e_pow.c ----
struct GdLog { double hi, lo; };
// log_inline: log(x) = hi + lo for the bits ix of a positive, normal(ised) x
GD_FN GdLog gd_pow_log(uint64_t ix)
{
    const double *P = gd_data.powlog_hdr, *A = P + 2;
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double z = gd_asdouble(iz), kd = (double)k;
    const double invc = GD_T.powlog_tab[3 * i], logc = GD_T.powlog_tab[3 * i + 1], logctail = GD_T.powlog_tab[3 * i + 2];
    const double t1 = __builtin_fma(kd, P[0], logc);
    const double r = __builtin_fma(z, invc, -1.0);
    const double ar = r * A[0];
    const double lo1 = __builtin_fma(kd, P[1], logctail);
    const double q12 = __builtin_fma(r, A[2], A[1]), q34 = __builtin_fma(r, A[4], A[3]);
    const double t2 = r + t1;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double lo2 = (t1 - t2) + r;
    const double q56 = __builtin_fma(r, A[6], A[5]);
    const double hi = t2 + ar2;
    const double q36 = __builtin_fma(q56, ar2, q34);
    const double lo4 = (t2 - hi) + ar2;
    const double q = __builtin_fma(ar2, q36, q12);
    double lo = lo1 + lo2;
    lo = lo + lo3;
    lo = lo + lo4;
    lo = __builtin_fma(ar3, q, lo);
    GdLog o;
    o.hi = hi + lo;
    o.lo = (hi - o.hi) + lo;
    return o;
}

// exp_inline(y * log x): the second half of pow for a finite y inside [2^-65, 2^63) in magnitude
GD_FN double gd_pow_exp(const GdLog &L, double y, uint64_t sign_bias)
{
    const double ehi = y * L.hi;
    double elo = __builtin_fma(L.hi, y, -ehi);
    elo = __builtin_fma(y, L.lo, elo);
    uint32_t abstop = gd_top12(ehi) & 0x7ff;
    if (GD_UNLIKELY(abstop - 0x3c9 >= 0x3f)) {
        if (abstop - 0x3c9 >= 0x80000000u) { const double one = 1.0 + ehi; return sign_bias ? -one : one; }
        if (abstop >= 0x409) {
            const double mag = (gd_asuint64(ehi) >> 63) ? 0.0 : gd_asdouble(GD_INF);
            return sign_bias ? -mag : mag;
        }
        abstop = 0;
    }
    return gd_exp_core<true>(ehi, elo, sign_bias, abstop);
}

// 0: y is not an integer, 1: odd, 2: even
GD_FN int gd_checkint(uint64_t iy)
{
    const int e = (int)((iy >> 52) & 0x7ff);
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ull << (0x3ff + 52 - e))) return 1;
    return 2;
}
GD_FN bool gd_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * GD_INF - 1; }

// everything __pow does before log_inline for arguments outside the main range; returns true when *res is the result,
// otherwise ix (normalised, sign removed) and sign_bias are ready for the main path
GD_FN bool gd_pow_special(double x, double y, uint64_t &ix, uint64_t &sign_bias, double *res)
{
    const uint64_t iy = gd_asuint64(y);
    uint32_t topx = gd_top12(x);
    const uint32_t topy = gd_top12(y);
    if (gd_zeroinfnan(iy)) {
        if (2 * iy == 0) { *res = 1.0; return true; }
        if (ix == GD_ONE) { *res = 1.0; return true; }
        if (2 * ix > 2 * GD_INF || 2 * iy > 2 * GD_INF) { *res = x + y; return true; }
        if (2 * ix == 2 * GD_ONE) { *res = 1.0; return true; }
        if ((2 * ix < 2 * GD_ONE) == !(iy >> 63)) { *res = 0.0; return true; }
        *res = y * y; return true;
    }
    if (gd_zeroinfnan(ix)) {
        double x2 = x * x;
        if ((ix >> 63) && gd_checkint(iy) == 1) x2 = -x2;
        *res = (iy >> 63) ? 1 / x2 : x2; return true;
    }
    if (ix >> 63) {                                          // finite x < 0
        const int yint = gd_checkint(iy);
        if (yint == 0) { *res = (x - x) / 0.0; return true; }
        if (yint == 1) sign_bias = 0x800ull << 7;
        ix &= 0x7fffffffffffffffull; topx &= 0x7ff;
    }
    if ((topy & 0x7ff) - 0x3be >= 0x80) {
        if (ix == GD_ONE) { *res = 1.0; return true; }
        if ((topy & 0x7ff) < 0x3be) { *res = ix > GD_ONE ? 1.0 + y : 1.0 - y; return true; }
        *res = ((ix > GD_ONE) == (topy < 0x800)) ? gd_asdouble(GD_INF) : 0.0; return true;
    }
    if (topx == 0) {                                         // subnormal x
        ix = gd_asuint64(x * 0x1p52);
        ix &= 0x7fffffffffffffffull;
        ix -= 52ull << 52;
    }
    return false;
}

GD_FN double gd_pow(double x, double y)
{
    uint64_t ix = gd_asuint64(x), sign_bias = 0;
    const uint32_t topx = gd_top12(x), topy = gd_top12(y);
    if (GD_UNLIKELY(topx - 0x001 >= 0x7ff - 0x001 || (topy & 0x7ff) - 0x3be >= 0x80)) {
        double res;
        if (gd_pow_special(x, y, ix, sign_bias, &res)) return res;
    }
    const GdLog L = gd_pow_log(ix);
    return gd_pow_exp(L, y, sign_bias);
}
