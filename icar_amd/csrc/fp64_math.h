// icar_amd/csrc/fp64_math.h -- FP64 log / exp polynomials of rounds 1-3.  NOT on the product path any more: the REAL(4)
// transcendentals are glibc's float functions (glibc_flt32.h, round 3) and the DOUBLE PRECISION ones glibc's double functions
// (glibc_dbl64.h, round 4).  What is still used is `DK` / `d_consts()`, the by-reference constant block the level code's helper
// signatures carry; the routines below are kept as the measured alternative the docs refer to (profiles/r04_steps.md).
#pragma once
#include <hip/hip_runtime.h>

// (Round 1 measured two variants of these routines on k_thompson_pack and dropped both -- DESIGN.md section 3: Horner
// coefficients fed from SGPR pairs, -12 % VALU instructions at an unchanged cycle count; an even/odd or Estrin split of the
// exp polynomial, no gain.  The kernel waits on the dependent-issue latency of its instruction stream, not on these chains.)
// Natural log of a positive finite double in ~38 instructions (ocml's log(double) is ~95: it carries a double-double
// result that a value about to be rounded to REAL(4) does not need).  Classic reduction x = 2^k m, m in [sqrt(1/2),
// sqrt(2)), s = f/(2+f) with f = m-1, log(m) = f - (f^2/2 - s (f^2/2 + R(s^2))) with the 7-term minimax R of
// W. Kahan / fdlibm e_log.c (error bound 2^-58.45); the quotient is a refined v_rcp_f64.  Measured against long
// double on 2e7 random REAL(4) arguments: max error 0.74 ulp of the double, no REAL(4) rounding differing from the
// exactly rounded one.
__device__ __forceinline__ double d_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);             // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m * 2.0 : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// exp of a double in ~20 instructions: x = k ln2 + r, |r| <= 0.347, degree-13 Horner, v_ldexp_f64 (saturates to 0 / inf).
// Against long double on 2e7 arguments in [-700, 700]: max error 0.87 ulp of the double.
__device__ __forceinline__ double d_exp(double x)
{
    const double invln2 = 1.44269504088896338700e+00, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double k = rint(x * invln2);
    double r = fma(-k, ln2_hi, x);
    r = fma(-k, ln2_lo, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);    p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);    p = fma(p, r, 1.0 / 40320.0);    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);       p = fma(p, r, 1.0 / 120.0);      p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);         p = fma(p, r, 0.5);                 p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}


// ---- the same two functions with their polynomial coefficients RESIDENT in scalar registers ----------------------------
// clang materialises the FP64 literal of every Horner step where it is used (two v_mov_b32 into a VGPR pair + v_fmac_f64,
// or two s_mov_b32 + v_fma_f64): three instructions per step.  A kernel that evaluates ~100 of these per thread loads the
// coefficients once (d_consts: opaque to the optimiser, so they are neither re-materialised nor folded) and every step is
// the one v_fma_f64 with an SGPR-pair addend.  Same operations, same order, same results as d_log / d_exp.
struct DK { double e[10], lg[6], ln2_hi, ln2_lo, invln2, e13, lg7, sqrt_half; };
__device__ __forceinline__ double d_opaque_sgpr(double c)
{
    unsigned lo = (unsigned)__double_as_longlong(c), hi = (unsigned)((unsigned long long)__double_as_longlong(c) >> 32);
    asm volatile("" : "+s"(lo));
    asm volatile("" : "+s"(hi));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ DK d_consts()
{
    DK K;
    const double e[10] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                          1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0};
    const double lg[6] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                          2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01};
    for (int i = 0; i < 10; ++i) K.e[i] = d_opaque_sgpr(e[i]);
    for (int i = 0; i < 6; ++i) K.lg[i] = d_opaque_sgpr(lg[i]);
    K.ln2_hi = d_opaque_sgpr(6.93147180369123816490e-01);
    K.ln2_lo = d_opaque_sgpr(1.90821492927058770002e-10);
    K.invln2 = d_opaque_sgpr(1.44269504088896338700e+00);
    K.e13 = d_opaque_sgpr(1.0 / 6227020800.0);
    K.lg7 = d_opaque_sgpr(1.479819860511658591e-01);
    K.sqrt_half = d_opaque_sgpr(0.70710678118654752440);
    return K;
}
__device__ __forceinline__ double d_log_k(const DK &K, double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < K.sqrt_half;
    m = lo ? m * 2.0 : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    // R(z) in two interleaved Horner chains; the steps whose addend is an SGPR pair are written as instructions, in ONE
    // statement each chain pair (the compiler pads every asm statement that consumes the result of the previous one with
    // an s_nop it cannot prove unnecessary; dependent v_fma_f64 need none)
    double ta = fma(w, K.lg[5], K.lg[3]), tb = fma(w, K.lg7, K.lg[4]);
    // (s_nop 1: gfx94x / gfx950 need two wait states between a VALU write of an SGPR -- the v_readlane_b32 that restores a
    // spilled coefficient -- and a VALU read of it; the compiler's hazard recogniser does not look inside an asm statement)
    asm("s_nop 1\n\tv_fma_f64 %0, %2, %0, %3\n\tv_fma_f64 %1, %2, %1, %4\n\tv_fma_f64 %1, %2, %1, %5"
        : "+v"(ta), "+v"(tb) : "v"(w), "s"(K.lg[1]), "s"(K.lg[2]), "s"(K.lg[0]));
    const double t1 = w * ta, t2 = z * tb;
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return fma(dk, K.ln2_hi, -((hfsq - fma(s, hfsq + R, dk * K.ln2_lo)) - f));   // k ln2_hi is exact (32 trailing zero bits): the value of d_log's mul + sub
}
__device__ __forceinline__ double d_exp_k(const DK &K, double x)
{
    const double k = rint(x * K.invln2);
    double r = fma(-k, K.ln2_hi, x);
    r = fma(-k, K.ln2_lo, r);
    double p = fma(r, K.e13, K.e[0]);
    asm("s_nop 1\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %5\n\t"
        "v_fma_f64 %0, %0, %1, %6\n\tv_fma_f64 %0, %0, %1, %7\n\tv_fma_f64 %0, %0, %1, %8\n\tv_fma_f64 %0, %0, %1, %9\n\t"
        "v_fma_f64 %0, %0, %1, %10\n\tv_fma_f64 %0, %0, %1, 0.5\n\tv_fma_f64 %0, %0, %1, 1.0\n\tv_fma_f64 %0, %0, %1, 1.0"
        : "+v"(p) : "v"(r), "s"(K.e[1]), "s"(K.e[2]), "s"(K.e[3]), "s"(K.e[4]), "s"(K.e[5]), "s"(K.e[6]), "s"(K.e[7]),
          "s"(K.e[8]), "s"(K.e[9]));
    return __builtin_amdgcn_ldexp(p, (int)k);
}
