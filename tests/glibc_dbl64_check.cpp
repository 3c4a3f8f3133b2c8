// tests/glibc_dbl64_check.cpp -- CPU check of icar_amd/csrc/glibc_dbl64.h (the device's DOUBLE PRECISION exp / log / pow) against
// the host C library, value by value.  Built and run by tests/test_glibc_dbl64_host.py:
//     g++ -O2 -mfma -ffp-contract=off -fopenmp glibc_dbl64_check.cpp -o ... ;  ./check <n>
// <n> random arguments per class: exp over [-750, 715], around 0 and over all bit patterns; log over all positive binades, densely
// around 1 and over all bit patterns; pow with positive bases over 10^-40 .. 10^40 and exponents in [-12, 12] (how the
// microphysics uses it), bases and exponents over all bit patterns, and a grid of special values.  Prints
// "<name> <tested> <mismatches>" per class.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define GD_FN static inline
#define GD_TABLE static const
#include "../icar_amd/csrc/glibc_dbl64.h"

static inline bool same(double a, double b)
{
    if (std::isnan(a) && std::isnan(b)) return true;
    return gd_asuint64(a) == gd_asuint64(b);
}
static inline uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static inline double unif(uint64_t &s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }

template <class GEN, class F, class G>
static void run1(const char *name, uint64_t n, GEN gen, F mine, G ref)
{
    uint64_t bad = 0; double fx = 0; bool have = false;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < (int64_t)n; ++t) {
        uint64_t s = 0x1234567ull + 0x9e3779b97f4a7c15ull * (uint64_t)t;
        const double x = gen(s);
        if (!same(mine(x), ref(x))) {
            ++bad;
#pragma omp critical
            if (!have) { have = true; fx = x; }
        }
    }
    printf("%s %llu %llu", name, (unsigned long long)n, (unsigned long long)bad);
    if (have) printf("  first: x=%a mine=%a ref=%a", fx, mine(fx), ref(fx));
    printf("\n");
}
template <class GEN>
static void run2(const char *name, uint64_t n, GEN gen)
{
    uint64_t bad = 0; double fx = 0, fy = 0; bool have = false;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < (int64_t)n; ++t) {
        uint64_t s = 0x7654321ull + 0x9e3779b97f4a7c15ull * (uint64_t)t;
        double x, y; gen(s, x, y);
        if (!same(gd_pow(x, y), pow(x, y))) {
            ++bad;
#pragma omp critical
            if (!have) { have = true; fx = x; fy = y; }
        }
    }
    printf("%s %llu %llu", name, (unsigned long long)n, (unsigned long long)bad);
    if (have) printf("  first: x=%a y=%a mine=%a ref=%a", fx, fy, gd_pow(fx, fy), pow(fx, fy));
    printf("\n");
}

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    auto e = [](double x) { return gd_exp(x); }; auto E = [](double x) { return exp(x); };
    auto l = [](double x) { return gd_log(x); }; auto L = [](double x) { return log(x); };
    run1("exp_range", n, [](uint64_t &s) { return -750.0 + 1465.0 * unif(s); }, e, E);
    run1("exp_small", n, [](uint64_t &s) { return (unif(s) - 0.5) * std::ldexp(1.0, -(int)(splitmix(s) % 70)); }, e, E);
    run1("exp_bits", n, [](uint64_t &s) { return gd_asdouble(splitmix(s)); }, e, E);
    run1("log_binades", n, [](uint64_t &s) { return std::ldexp(1.0 + unif(s), (int)(splitmix(s) % 2098) - 1074); }, l, L);
    run1("log_near1", n, [](uint64_t &s) { return 1.0 + (unif(s) - 0.5) * std::ldexp(1.0, -(int)(splitmix(s) % 50)); }, l, L);
    run1("log_bits", n, [](uint64_t &s) { return gd_asdouble(splitmix(s)); }, l, L);
    run2("pow_physics", n, [](uint64_t &s, double &x, double &y) { x = std::pow(10.0, -40.0 + 80.0 * unif(s)); y = -12.0 + 24.0 * unif(s); });
    run2("pow_quarters", n, [](uint64_t &s, double &x, double &y) { x = std::pow(10.0, -20.0 + 40.0 * unif(s)); y = 0.25 * (double)((int)(splitmix(s) % 97) - 48); });
    run2("pow_one", n, [](uint64_t &s, double &x, double &y) { x = gd_asdouble(splitmix(s)); y = 1.0; });       // and pow(x, 1) IS x:
    { uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad)
      for (int64_t t = 0; t < (int64_t)n; ++t) { uint64_t s = 0x51ull + 0x9e3779b97f4a7c15ull * (uint64_t)t; const double x = gd_asdouble(splitmix(s)); if (!same(pow(x, 1.0), x)) ++bad; }
      printf("libm_pow_one_is_x %llu %llu\n", (unsigned long long)n, (unsigned long long)bad); }
    run2("pow_bits", n, [](uint64_t &s, double &x, double &y) { x = gd_asdouble(splitmix(s)); y = gd_asdouble(splitmix(s)); });
    run2("pow_posbits", n, [](uint64_t &s, double &x, double &y) { x = gd_asdouble(splitmix(s) >> 1); y = (unif(s) - 0.5) * std::ldexp(1.0, (int)(splitmix(s) % 24) - 10); });
    run2("pow_extreme", n, [](uint64_t &s, double &x, double &y) { x = std::ldexp(1.0 + unif(s), (int)(splitmix(s) % 2098) - 1074); y = (unif(s) - 0.5) * 2200.0 / std::fmax(1.0, std::fabs(std::log2(x))); });
    // grid of special values
    const double sp[] = {0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 3.0, -3.0, 1e-320, -1e-320, 0x1p-1022, 0x1p1023, -0x1p1023, INFINITY, -INFINITY, NAN,
                         0x1p-70, -0x1p-70, 0x1p70, 1.5, 2.5, 1024.0, -1075.0, 0x1.fffffffffffffp-1, 0x1.0000000000001p0};
    uint64_t bad = 0, cnt = 0;
    for (double x : sp) for (double y : sp) { ++cnt; if (!same(gd_pow(x, y), pow(x, y))) { if (!bad) printf("  special first: x=%a y=%a mine=%a ref=%a\n", x, y, gd_pow(x, y), pow(x, y)); ++bad; } }
    for (double x : sp) { cnt += 2; if (!same(gd_exp(x), exp(x))) { printf("  exp special x=%a mine=%a ref=%a\n", x, gd_exp(x), exp(x)); ++bad; } if (!same(gd_log(x), log(x))) { printf("  log special x=%a mine=%a ref=%a\n", x, gd_log(x), log(x)); ++bad; } }
    printf("special %llu %llu\n", (unsigned long long)cnt, (unsigned long long)bad);
    return 0;
}
