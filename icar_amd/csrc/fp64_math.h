// icar_amd/csrc/fp64_math.h -- FP64 log / exp for REAL(4) transcendentals evaluated in double and rounded once.
#pragma once
#include <hip/hip_runtime.h>

// fma(a, b, C) / fma(-a, C, c) with a compile-time constant C held in an SGPR pair.  Left to itself the compiler turns a
// Horner step into `v_mov_b32 x2 ; v_fmac_f64` (the VOP2 form wants the addend in the destination VGPR, and a 64-bit
// literal cannot be an operand on gfx9): three VALU issues instead of one.  The "s" constraint materialises C with two
// s_mov_b32 on the scalar pipe instead.  Same operation, same rounding.
// Measured on k_thompson_pack (profiles/micro, -DICAR_FMA_SC=1): VALU instructions -12 % (1.36e9 -> 1.20e9 per launch),
// SALU +36 %, bit-identical results -- and the same number of cycles: that kernel is bound by the dependent-issue
// latency of each wave's instruction stream at 4 waves per SIMD, not by VALU throughput.
// Off by default: no gain, and plain fma() leaves the scheduling to the compiler.
#ifndef ICAR_FMA_SC
#define ICAR_FMA_SC 0
#endif
// ICAR_EXP_SPLIT (A/B builds, d_exp): 1 = even/odd half chains (depth 15 instead of 18, max error 1.04 ulp of the double),
// 2 = Estrin (depth 10, 2.2 ulp); neither changes a REAL(4) rounding on 3e7 arguments.  k_thompson_pack: 0 -> 2.28 ms,
// 1 -> 2.28-2.30 ms, 2 -> 2.43 ms (more instructions): the chain length of one exp is not what the kernel waits for.
#ifndef ICAR_EXP_SPLIT
#define ICAR_EXP_SPLIT 0
#endif
__device__ __forceinline__ double fma_sc(double a, double b, double C)
{
#if ICAR_FMA_SC
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(C));
    return r;
#else
    return fma(a, b, C);
#endif
}
__device__ __forceinline__ double fnma_sc(double a, double C, double c)       // fma(-a, C, c)
{
#if ICAR_FMA_SC
    double r;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "s"(C), "v"(c));
    return r;
#else
    return fma(-a, C, c);
#endif
}

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
    const double t1 = w * fma_sc(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_sc(w, fma_sc(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// exp of a double in ~20 instructions: x = k ln2 + r, |r| <= 0.347, degree-13 Horner, v_ldexp_f64 (saturates to 0 / inf).
// Against long double on 2e7 arguments in [-700, 700]: max error 0.87 ulp of the double.
__device__ __forceinline__ double d_exp(double x)
{
    const double invln2 = 1.44269504088896338700e+00, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double k = rint(x * invln2);
    double r = fnma_sc(k, ln2_hi, x);
    r = fnma_sc(k, ln2_lo, r);
#if ICAR_EXP_SPLIT == 2
    // Estrin: exp(r) = sum c_i r^i, c_i = 1/i!, pairs -> quads -> octets; dependent depth 5 instead of 13
    const double z = r * r, z2 = z * z, z4 = z2 * z2;
    const double p01 = fma(r, 1.0, 1.0), p23 = fma(r, 1.0 / 6.0, 0.5), p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    const double p67 = fma(r, 1.0 / 5040.0, 1.0 / 720.0), p89 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
    const double pab = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0), pcd = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double q0 = fma(z, p23, p01), q1 = fma(z, p67, p45), q2 = fma(z, pab, p89);
    const double h0 = fma(z2, q1, q0), h1 = fma(z2, pcd, q2);
    double p = fma(z4, h1, h0);
#elif ICAR_EXP_SPLIT == 1
    // even / odd halves: exp(r) = 1 + r + z (E(z) + r O(z)), two 5-step chains instead of one of 13
    const double z = r * r;
    double e = 1.0 / 479001600.0, o = 1.0 / 6227020800.0;
    e = fma(e, z, 1.0 / 3628800.0); o = fma(o, z, 1.0 / 39916800.0);   // both operands constants: left to the compiler
    e = fma_sc(e, z, 1.0 / 40320.0);   o = fma_sc(o, z, 1.0 / 362880.0);
    e = fma_sc(e, z, 1.0 / 720.0);     o = fma_sc(o, z, 1.0 / 5040.0);
    e = fma_sc(e, z, 1.0 / 24.0);      o = fma_sc(o, z, 1.0 / 120.0);
    e = fma(e, z, 0.5);             o = fma_sc(o, z, 1.0 / 6.0);
    double p = fma(z, fma(r, o, e), r) + 1.0;
#else
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);    p = fma_sc(p, r, 1.0 / 39916800.0); p = fma_sc(p, r, 1.0 / 3628800.0);
    p = fma_sc(p, r, 1.0 / 362880.0);    p = fma_sc(p, r, 1.0 / 40320.0);    p = fma_sc(p, r, 1.0 / 5040.0);
    p = fma_sc(p, r, 1.0 / 720.0);       p = fma_sc(p, r, 1.0 / 120.0);      p = fma_sc(p, r, 1.0 / 24.0);
    p = fma_sc(p, r, 1.0 / 6.0);         p = fma(p, r, 0.5);                 p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
#endif
    return __builtin_amdgcn_ldexp(p, (int)k);
}

