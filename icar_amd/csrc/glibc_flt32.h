// icar_amd/csrc/glibc_flt32.h -- expf / logf / log10f / powf / atanf exactly as the compiled reference evaluates them.
// Origin and licence: restated from the GNU C Library 2.35 (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_powf.c with math_config.h's
// data tables: Copyright (C) Free Software Foundation, Inc. / Arm Ltd., "optimized routines"; e_log10f.c, s_atanf.c: Copyright (C)
// 1993 Sun Microsystems, Inc. (fdlibm), as distributed in glibc), GNU Lesser General Public License 2.1 or later (the Arm originals
// also MIT; fdlibm: "permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is
// preserved").  The algorithms and constants are theirs; this file is a derived work under the same terms.
//
// The reference is Fortran; flang lowers REAL(4) exp / log / log10 / x**y / atan to the C library's expf / logf / log10f /
// powf / atanf, and the image's C library is glibc 2.35 (not vendored in /root/reference).  Its float functions are the
// published ARM "optimized routines" algorithms (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_powf.c: one table look-up and a
// low-degree polynomial in double precision, rounded once to float) plus the fdlibm e_log10f.c / s_atanf.c.  They are restated
// here for the device so that the microphysics, the exner function and the linear-wind stability see bit for bit the
// transcendental the compiled reference sees -- the anchor of every HIP-vs-oracle parity test is then the reference's own
// math (oracle "mode 0" = the host's libm), not a definition of ours.
//
// x86-64 glibc selects its FMA builds of e_expf / e_logf / e_powf at load time on any CPU with AVX2 + FMA (every host an
// MI355X sits in); those contract a*b+c exactly where written below as fma().  The operation sequence was read off the
// image's own libm.so.6 and is checked, value by value, against that libm on the host: tests/test_glibc_flt32_host.py
// compiles this header for the CPU (every REAL(4) argument of expf / logf / log10f / atanf, 10^9 argument pairs of powf) and
// tests/test_gpu_glibc_math.py runs it on the device.  Tables and coefficients are glibc's published constants (data).
//
// Cost per call (FP64 VALU instructions; the FP64 polynomial d_exp / d_log of rounds 1-2 they replaced: 20 / 38):
// expf 8, logf 7, powf 17 (log2 part 9, shared by all powers of one base; exp2 part 8).
#pragma once
#include <stdint.h>

// Shape of every function: the main path is evaluated unconditionally (its table indices are masked, so any bit pattern is a
// safe argument) and ONE never-taken branch replaces the value for the arguments glibc treats separately (zero, negative,
// subnormal, inf, NaN, overflow): a wave of the microphysics pays no exec-mask bookkeeping per special case.
//
// Tables: GF_LDS_TABLES defined before the include -> the three tables (768 B) live in LDS (`gf_lds`; every kernel of that
// translation unit that calls these functions runs gf_lds_init() + a barrier first): a ds_read_b64 / b128 per call instead of
// a dependent global load.  Otherwise they are read from constant global memory.
#ifndef GF_FN
#define GF_FN __device__ __forceinline__
#define GF_TABLE __device__ const
#endif
#ifndef GF_UNLIKELY
#define GF_UNLIKELY(c) __builtin_expect(!!(c), 0)
#endif

GF_FN uint32_t gf_asuint(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
GF_FN float gf_asfloat(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
GF_FN uint64_t gf_asuint64(double f) { uint64_t u; __builtin_memcpy(&u, &f, 8); return u; }
GF_FN double gf_asdouble(uint64_t u) { double f; __builtin_memcpy(&f, &u, 8); return f; }

// __exp2f_data.tab: T[i] = bits(2^(i/32)) - (i << 47)                                              (e_exp2f_data.c)
struct GfTab { uint64_t exp2[32]; double logt[16][2]; double powt[16][2]; };
GF_TABLE GfTab gf_tab = {{
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540},
// __logf_data.tab {invc, logc}                                                                      (e_logf_data.c)
   {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1.0000000000000p+0, 0x0.0p+0},               {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},  {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}},
// __powf_log2_data.tab {invc, logc} (logc in units of log2)                                         (e_powf_log2_data.c)
   {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1.0000000000000p+0, 0x0.0p+0},               {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}}};
#ifdef GF_LDS_TABLES
__shared__ GfTab gf_lds;
// every thread of the block calls this once at the top of the kernel (nthreads = threads of the block); ends with a barrier
__device__ __forceinline__ void gf_lds_init(int tid, int nthreads)
{
    const uint64_t *src = (const uint64_t *)&gf_tab; uint64_t *dst = (uint64_t *)&gf_lds;
    for (int t = tid; t < (int)(sizeof(GfTab) / 8); t += nthreads) dst[t] = src[t];
    __syncthreads();
}
#define GF_T gf_lds
#else
#define GF_T gf_tab
#endif

// ---- expf (e_expf.c, FMA build) ----------------------------------------------------------------------------------------
GF_FN double gf_scale(uint32_t ki_lo, uint32_t sign_bias)
{   // asdouble(T[ki % 32] + ((ki + sign_bias) << 47)): the shifted term has a zero low word, so only the high word adds
    const uint64_t t = GF_T.exp2[ki_lo & 31];
    const uint32_t hi = (uint32_t)(t >> 32) + ((ki_lo + sign_bias) << 15);
    return gf_asdouble(((uint64_t)hi << 32) | (uint32_t)t);
}
GF_FN float gf_expf_special(float x, float main_value)
{   // |x| >= 88 or NaN
    if (gf_asuint(x) == 0xff800000u) return 0.0f;              // exp(-inf)
    if (((gf_asuint(x) >> 20) & 0x7ff) >= 0x7f8) return x + x;  // +inf, NaN
    if (x > 0x1.62e42ep6f) return __builtin_inff();            // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                       // underflow
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;                  // __math_may_uflowf: (0x1.4p-75f)**2 rounded
    return main_value;                                         // -0x1.9d1d9ep6 <= x <= -88: a subnormal result, from the formula
}
GF_FN float gf_expf(float x)
{
    const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52,                           // 32 / ln 2
                 C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;     // __exp2f_data.poly
    const double xd = (double)x;
    double kd = __builtin_fma(InvLn2N, xd, Shift);            // z + Shift in ONE rounding in the FMA build
    const uint32_t ki = (uint32_t)gf_asuint64(kd);
    kd -= Shift;
    // glibc's poly_scaled is poly / 32**(3-i): with r / 32 (exact) the unscaled coefficients give the same bits in every step
    const double r = __builtin_fma(InvLn2N, xd, -kd) * 0x1p-5;
    const double s = gf_scale(ki, 0);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    float res = (float)(y * s);
    if (GF_UNLIKELY(((gf_asuint(x) >> 20) & 0x7ff) > 0x42a)) res = gf_expf_special(x, res);
    return res;
}

// ---- logf (e_logf.c, FMA build) ----------------------------------------------------------------------------------------
GF_FN float gf_logf_main(uint32_t ix)
{   // a positive normal x as its bit pattern (exponent possibly below the normal range after the subnormal shift); log(1) = +0
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);             // (0x1ff << 23)
    const double invc = GF_T.logt[i][0], logc = GF_T.logt[i][1];
    const double z = (double)gf_asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    return (float)__builtin_fma(y, r2, y0 + r);
}
GF_FN float gf_logf_special(float x)
{   // x < 0x1p-126 or inf or nan
    const uint32_t ix = gf_asuint(x);
    if (ix * 2 == 0) return -__builtin_inff();                 // log(+-0)
    if (ix == 0x7f800000u) return x;                           // log(inf)
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);   // x < 0, NaN
    return gf_logf_main(gf_asuint(x * 0x1p23f) - (23u << 23));  // subnormal: normalise
}
GF_FN float gf_logf(float x)
{
    const uint32_t ix = gf_asuint(x);
    float res = gf_logf_main(ix);
    if (GF_UNLIKELY(ix - 0x00800000u >= 0x7f800000u - 0x00800000u)) res = gf_logf_special(x);
    return res;
}

// ---- log10f (e_log10f.c of glibc <= 2.39: fdlibm, plain float arithmetic around logf) -------------------------------------
GF_FN float gf_log10f_main(int32_t hx, int32_t k)
{
    const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    k += (hx >> 23) - 127;
    const int32_t i = (int32_t)((uint32_t)k >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    const float z = y * log10_2lo + ivln10 * gf_logf_main((uint32_t)hx);      // the argument is in [0.5, 2): logf's main path
    return z + y * log10_2hi;
}
GF_FN float gf_log10f_special(float x)
{
    const float two25 = 3.3554432000e+07f;
    int32_t hx = (int32_t)gf_asuint(x);
    if (hx >= 0x7f800000) return x + x;
    if ((hx & 0x7fffffff) == 0) return -two25 / __builtin_fabsf(x);         // log(+-0) = -inf
    if (hx < 0) return (x - x) / (x - x);                      // log(-#) = NaN
    x *= two25;                                                // subnormal: scale up
    return gf_log10f_main((int32_t)gf_asuint(x), -25);
}
GF_FN float gf_log10f(float x)
{
    const uint32_t ix = gf_asuint(x);
    float res = gf_log10f_main((int32_t)ix, 0);
    if (GF_UNLIKELY(ix - 0x00800000u >= 0x7f800000u - 0x00800000u)) res = gf_log10f_special(x);
    return res;
}

// ---- powf (e_powf.c, FMA build) ----------------------------------------------------------------------------------------
// log2(x) of a normal positive x (as its bit pattern, exponent possibly below the normal range after the subnormal shift) to
// ~2^-40 relative: the part of powf that depends on the base alone.  x ** y for several y of one base shares it (the same bits).
GF_FN double gf_powf_log2(uint32_t ix)
{
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp+0;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = GF_T.powt[i][0], logc = GF_T.powt[i][1];
    const double z = (double)gf_asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double y = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r2 = r * r;
    double q = __builtin_fma(A4, r, y0);
    const double r4 = r2 * r2;
    q = __builtin_fma(p, r2, q);
    return __builtin_fma(y, r4, q);
}

// 2^xd for |xd| < 150 (sign_bias = 1 << 16 negates the result: a negative base to an odd integer power)
GF_FN float gf_powf_exp2(double xd, uint32_t sign_bias)
{
    const double Shift = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;   // 0x1.8p52 / 32
    double kd = xd + Shift;
    const uint32_t ki = (uint32_t)gf_asuint64(kd);
    kd -= Shift;
    const double r = xd - kd;
    const double s = gf_scale(ki, sign_bias);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    return (float)(y * s);
}

// |y log2 x| >= 126: powf's overflow / underflow rules apply
GF_FN bool gf_powf_big(double ylogx) { return (((uint32_t)(gf_asuint64(ylogx) >> 32) >> 15) & 0xffff) >= (0x405f800000000000ull >> 47); }
GF_FN float gf_powf_finish(double ylogx, uint32_t sign_bias)
{
    if (gf_powf_big(ylogx)) {
        const float sgn = sign_bias ? -1.0f : 1.0f;
        if (ylogx > 0x1.fffffffd1d571p+6) return sgn * __builtin_inff();               // > 127.99999995
        if (ylogx <= -150.0) return sgn * 0.0f;
        if (ylogx < -149.0) return sgn * 0x1p-149f;                                    // __math_may_uflowf
    }
    return gf_powf_exp2(ylogx, sign_bias);
}

GF_FN int gf_checkint(uint32_t iy)
{   // 0: not an integer, 1: odd, 2: even
    const int e = (iy >> 23) & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

// the whole of glibc's powf, every case
GF_FN float gf_powf_full(float x, float y)
{
    uint32_t sign_bias = 0;
    uint32_t ix = gf_asuint(x);
    const uint32_t iy = gf_asuint(y);
    const bool y_zeroinfnan = 2 * iy - 1 >= 2u * 0x7f800000u - 1;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || y_zeroinfnan) {
        if (y_zeroinfnan) {
            if (2 * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;      // |x| < 1 && y == inf, |x| > 1 && y == -inf
            return y * y;
        }
        if (2 * ix - 1 >= 2u * 0x7f800000u - 1) {                                      // x is +-0, +-inf or NaN
            float x2 = x * x;
            if ((ix & 0x80000000u) && gf_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;                                // 1 / +-0 = +-inf
        }
        if (ix & 0x80000000u) {                                                        // finite x < 0
            const int yint = gf_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 1u << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                                        // subnormal x: normalise
            ix = gf_asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    return gf_powf_finish((double)y * gf_powf_log2(ix), sign_bias);
}

// x ** y when l2 = gf_powf_log2(bits of x) is at hand (several powers of one base): the main path from l2, everything glibc
// treats separately -- base not a positive normal number, y zero / inf / NaN, |y log2 x| >= 126 -- behind one branch
GF_FN float gf_powf_from_log2(float x, double l2, float y)
{
    const uint32_t ix = gf_asuint(x), iy = gf_asuint(y);
    const double ylogx = (double)y * l2;
    float res = gf_powf_exp2(ylogx, 0);
    const bool special = (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) | (2 * iy - 1 >= 2u * 0x7f800000u - 1) | gf_powf_big(ylogx);
    if (GF_UNLIKELY(special)) res = gf_powf_full(x, y);
    return res;
}
GF_FN float gf_powf(float x, float y) { return gf_powf_from_log2(x, gf_powf_log2(gf_asuint(x)), y); }

// ---- atanf (s_atanf.c: fdlibm, plain float arithmetic; no FMA build exists for it) ------------------------------------------
GF_FN float gf_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)gf_asuint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                    // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                     // |x| < 0.4375
        if (ix < 0x31000000) return x;                         // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {                                 // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }       // 7/16 <= |x| < 11/16
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                               // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }        // |x| < 2.4375
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
