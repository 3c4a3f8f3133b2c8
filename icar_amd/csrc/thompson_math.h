// icar_amd/csrc/thompson_math.h -- the transcendentals and decade-table indices of the Thompson level code as device
// functions: DOUBLE PRECISION x**y / log / exp and REAL(4) powf / expf / log10f / atanf = the C library's (glibc_dbl64.h,
// glibc_flt32.h), the shared-base forms, 10.**n with an integer exponent, the decade index with its fast path.
// Included by mp_thompson.hip (the product) and by tests/support/th_probe.hip, which evaluates these very functions on arrays of
// arguments for the parity tests (the library itself exports no probe).
#pragma once
#include "thompson_state.h"
#include "fp64_math.h"
#define GF_LDS_TABLES          // the look-up tables of expf / logf / powf in LDS: every kernel that uses them starts with th_lds_init()
#include "glibc_flt32.h"
#define GD_LDS_TABLES          // ... and those of the DOUBLE PRECISION pow / log / exp (7 KB): th_lds_init() = both
#include "glibc_dbl64.h"
__device__ __forceinline__ void th_lds_init(int tid, int nthreads) { gd_lds_init(tid, nthreads); gf_lds_init(tid, nthreads); }

namespace {
// DOUBLE PRECISION x**y, log, exp: the C library's pow / log / exp bit for bit (glibc_dbl64.h), which is what the compiled reference
// calls (round 4: until then exp(y log x) with FP64 polynomials of our own -- < 1 ulp of the double, and one float ulp away from
// the reference in ~1e-7 of the cells of a step).  pow is log_inline (a function of the base alone: the powers of one base share
// it, d_plog / d_pow_l -- the same bits as separate pow calls) followed by exp_inline.
// every kernel below holds `const DK K_ = d_consts();` (fp64_math.h) for the REAL(4) helpers that still take it
#define d_exp(x) gd_exp(x)
#define d_log(x) gd_log(x)
#define d_pow(x, y) d_pow_k((x), (y))
#define d_pow_lx(L, x, y) d_pow_lx_k((L), (x), (y))
#define d_plog(x) gd_pow_log(gd_asuint64(x))          /* of a positive, normal DOUBLE PRECISION base */
#define d_powf(x, y) d_powf_k(K_, (x), (y))
#define d_pow_l(L, y) d_pow_l_k((L), (y))
#define d_powf_l(L, y) d_powf_l_k(K_, (L), (y))
#define d_pow10f(y) d_pow10f_k(K_, (y))
#define d_expf(x) d_expf_k(K_, (x))
#define d_log10f(x) d_log10f_k(K_, (x))
// x**y from L = log_inline(x) for the scheme's exponents (finite, 2^-65 <= |y| < 2^63, or zero)
__device__ __forceinline__ double d_pow_l_k(const GdLog &L, double y) { return (y == 0.0) ? 1.0 : gd_pow_exp(L, y, 0); }
// x**1 is x: glibc's pow errs by less than one ulp (0.52), and the only double within one ulp of x is x -- so the library itself
// returns x, bit for bit (checked on 1e9 bases in tests/glibc_dbl64_check.cpp, class pow_one).  The exponents mu_r + 1 and mu_g + 1
// of N0_r / N0_g are 1 with the default parameters: a quarter of a column's pow calls.  The test is wave-uniform (a parameter).
__device__ __forceinline__ double d_pow_k(double x, double y) { return (y == 1.0) ? x : gd_pow(x, y); }
__device__ __forceinline__ double d_pow_lx_k(const GdLog &L, double x, double y) { return (y == 1.0) ? x : d_pow_l_k(L, y); }
// REAL(4) x**y, exp, log10: the C library's powf / expf / log10f bit for bit (glibc_flt32.h), which is what the compiled
// reference calls.  powf is exp2(y * log2 x) with the log2 part a function of the base alone: powers of one base share it
// (PowBase; the same bits as separate powf calls, any base -- an unusual one takes powf itself).
__device__ __forceinline__ float d_powf_k(const DK &, float x, float y) { return gf_powf(x, y); }
struct PowBase { double l2; float x; };
__device__ __forceinline__ PowBase d_powf_base(float x) { PowBase b; b.x = x; b.l2 = gf_powf_log2(gf_asuint(x)); return b; }
__device__ __forceinline__ float d_powf_l_k(const DK &, const PowBase &b, float y) { return gf_powf_from_log2(b.x, b.l2, y); }
// 10.**y (REAL y): powf(10, y) with its log2 part, gf_powf_log2(bits of 10.0f), folded (tests/test_gpu_glibc_math.py op 9)
__device__ __forceinline__ float d_pow10f_k(const DK &, float y) { return gf_powf_from_log2(10.0f, 0x1.a934f0979b22dp+1, y); }
__device__ __forceinline__ float d_expf_k(const DK &, float x) { return gf_expf(x); }
__device__ __forceinline__ float d_log10f_k(const DK &, float x) { return gf_log10f(x); }


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
__device__ __forceinline__ int dec_index_f_slow(const DK &K_, float r, int n2)
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
__device__ __forceinline__ int dec_index_d_slow(const DK &K_, double r, int n2)
{
    const int nic = (int)lround(log10(r));
    int n = nic - 1;
    for (int nn = nic - 1; nn <= nic + 1; ++nn) {
        n = nn;
        if ((r / (double)powi10f(nn)) >= 1.0 && (r / (double)powi10f(nn)) < 10.0) break;
    }
    return (int)(r / (double)powi10f(n)) + 10 * (n - n2) - (n - n2);
}

/* The two routines above cost ~200 instructions per index (a logarithm, up to three trips of repeated squaring, a
 * reciprocal and two divisions each) and a level evaluates up to eight of them.  Their result is n = the decade D with
 * 10**D <= r < 10**(D+1) whenever r is not within rounding distance of a power of ten: the loop starts at nic-1 with
 * nic = nint(log10 r) in {D, D+1}, its test fails for D-1 and holds for D.  So: D from the hardware log2 (error ~1e-5
 * decades); if the fractional part of log10 r is at least 2e-4 away from 0 and 1 (the float powers of ten are within 1e-6
 * of the exact ones) the index is (int)(r / 10**D) + 9 (D - n2) with the SAME float 10**D (table filled by powi10f) and
 * the same IEEE division; otherwise (about 4 values in 10^4) the reference's loop runs. */
__device__ __forceinline__ bool dec_fast(const float *__restrict__ p10, float rf, int &D, float &p)
{
    const float t = __builtin_amdgcn_logf(rf) * 0.30102999566f;          /* v_log_f32 = log2 */
    const float fl = floorf(t), fr = t - fl;
    D = (int)fl;
    const bool ok = (fr > 2.e-4f) && (fr < 1.0f - 2.e-4f) && (D >= -TH_P10_OFF + 1) && (D <= TH_P10_N - TH_P10_OFF - 2) && (rf > 1.e-37f);
    p = p10[ok ? D + TH_P10_OFF : TH_P10_OFF];
    return ok;
}
__device__ __forceinline__ int dec_index_f_k(const DK &K_, const float *__restrict__ p10, float r, int n2)
{
    int D; float p;
    if (dec_fast(p10, r, D, p)) return (int)(r / p) + 10 * (D - n2) - (D - n2);
    return dec_index_f_slow(K_, r, n2);
}
__device__ __forceinline__ int dec_index_d_k(const DK &K_, const float *__restrict__ p10, double r, int n2)
{
    int D; float p;
    if (r < 1.e37 && dec_fast(p10, (float)r, D, p)) return (int)(r / (double)p) + 10 * (D - n2) - (D - n2);
    return dec_index_d_slow(K_, r, n2);
}

#define dec_index_f(T, r, n2) dec_index_f_k(K_, (T)->p10, (r), (n2))
#define dec_index_d(T, r, n2) dec_index_d_k(K_, (T)->p10, (r), (n2))
}  // namespace
