// tests/glibc_flt32_check.cpp -- CPU check of icar_amd/csrc/glibc_flt32.h (the device's expf / logf / log10f / powf / atanf)
// against the host C library, value by value.  Built and run by tests/test_glibc_flt32_host.py:
//     g++ -O2 -mfma -ffp-contract=off -fopenmp glibc_flt32_check.cpp -o ... ;  ./check <stride> <pairs>
// One-argument functions: every REAL(4) bit pattern whose index is a multiple of <stride> (1 = all 2^32).  powf: <pairs>
// random argument pairs drawn the way the microphysics uses it (positive bases over all binades, exponents in [-12, 12]) plus
// pairs over all finite bit patterns, plus a grid of special values.  Prints "<name> <tested> <mismatches>" per function.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#define GF_FN static inline
#define GF_TABLE static const
#include "../icar_amd/csrc/glibc_flt32.h"

static inline bool same(float a, float b)
{
    if (std::isnan(a) && std::isnan(b)) return true;
    return gf_asuint(a) == gf_asuint(b);
}
static inline uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <class F, class G>
static void sweep(const char *name, F mine, G ref, uint64_t stride)
{
    uint64_t bad = 0, n = 0; uint32_t first = 0; bool have = false;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (int64_t b = 0; b < (int64_t)1 << 32; b += (int64_t)stride) {
        const float x = gf_asfloat((uint32_t)b);
        ++n;
        if (!same(mine(x), ref(x))) {
            ++bad;
#pragma omp critical
            if (!have) { have = true; first = (uint32_t)b; }
        }
    }
    printf("%s %llu %llu", name, (unsigned long long)n, (unsigned long long)bad);
    if (have) { const float x = gf_asfloat(first); printf("  first: x=%a mine=%a ref=%a", x, mine(x), ref(x)); }
    printf("\n");
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    const uint64_t pairs = argc > 2 ? strtoull(argv[2], 0, 10) : 100000000ull;
    sweep("expf", gf_expf, [](float x) { return expf(x); }, stride);
    sweep("logf", gf_logf, [](float x) { return logf(x); }, stride);
    sweep("log10f", gf_log10f, [](float x) { return log10f(x); }, stride);
    sweep("atanf", gf_atanf, [](float x) { return atanf(x); }, stride);
    uint64_t bad = 0, n = 0; float fx = 0, fy = 0; bool have = false;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (int64_t t = 0; t < (int64_t)pairs; ++t) {
        uint64_t s = 0x1234567ull + (uint64_t)t * 0x2545f4914f6cdd1dull;
        const uint64_t a = splitmix(s), b = splitmix(s);
        float x, y;
        if (t & 1) {            // the microphysics' use: positive base from any binade, moderate exponent
            x = gf_asfloat((uint32_t)(a % 0x7f800000u));
            y = (float)((double)(b >> 11) / 9007199254740992.0 * 24.0 - 12.0);
        } else { x = gf_asfloat((uint32_t)a); y = gf_asfloat((uint32_t)b); }
        ++n;
        if (!same(gf_powf(x, y), powf(x, y))) {
            ++bad;
#pragma omp critical
            if (!have) { have = true; fx = x; fy = y; }
        }
    }
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.0f, -2.0f, 0.5f, -0.5f, 3.0f, -3.0f, 1e-45f, -1e-45f, 1e-40f, 1.17549435e-38f, 3.4028235e38f, -3.4028235e38f,
                        INFINITY, -INFINITY, NAN, 1.5f, -1.5f, 1e10f, -1e10f, 16777216.0f, 16777217.0f, 8388609.0f, -7.0f, 0.3333333f, 127.0f, 128.0f, -149.0f, -150.0f, 1e-5f};
    for (float x : sp) for (float y : sp) { ++n; if (!same(gf_powf(x, y), powf(x, y))) { ++bad; if (!have) { have = true; fx = x; fy = y; } } }
    // exponents that drive y log2 x through the overflow / underflow thresholds
    for (int e = -1600; e <= 1600; ++e) for (float x : {2.0f, 0.5f, 1.0000001f, 10.0f, 1e-40f}) {
        const float y = (float)e * 0.1f; ++n;
        if (!same(gf_powf(x, y), powf(x, y))) { ++bad; if (!have) { have = true; fx = x; fy = y; } }
    }
    printf("powf %llu %llu", (unsigned long long)n, (unsigned long long)bad);
    if (have) printf("  first: x=%a y=%a mine=%a ref=%a", fx, fy, gf_powf(fx, fy), powf(fx, fy));
    printf("\n");
    return 0;
}
