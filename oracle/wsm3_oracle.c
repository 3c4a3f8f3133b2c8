/* oracle/wsm3_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * WSM3 microphysics on the CPU: the column restatement of oracle/wsm3_column_oracle.h (src/physics/mp_wsm3.f90:218-903,
 * :951-1068, :1266-1505, each block citing its lines) compiled as plain C, driven like wsm3 (:74-216) drives wsm32D.
 * PINNED by execution: tests/test_oracle_wsm3.py compares it bit-for-bit with the unmodified mp_wsm3.f90 compiled into
 * oracle/_ref (constants of wsm3init and whole tiles over several steps).
 * Math mode (icar_oracle.c: orc_set_math_mode): 0 = libm expf/logf/powf as the compiled Fortran calls them, 1 = the FP64
 * function rounded once (what the HIP kernel evaluates).
 */
#include <math.h>
#include <stddef.h>
extern int g_math_mode;
static inline float o_expf(float x) { return g_math_mode ? (float)exp((double)x) : expf(x); }
static inline float o_logf(float x) { return g_math_mode ? (float)log((double)x) : logf(x); }
static inline float o_powf(float x, float y) { return g_math_mode ? (float)pow((double)x, (double)y) : powf(x, y); }
#define W3_FN static inline
#define W3_EXP(x) o_expf(x)
#define W3_LOG(x) o_logf(x)
#define W3_POW(x, y) o_powf(x, y)
#define W3_SQRT(x) sqrtf(x)
#define W3_MAXK 128
#define W3_HOST_INIT
#include "wsm3_column_oracle.h"

static wsm3_consts g_c;

/* out[0..41] in the order of struct wsm3_consts */
void orc_wsm3_init(float den0, float denr, float dens, float cl, float cpv, float *out)
{
    wsm3_init_consts(&g_c, den0, denr, dens, cl, cpv);
    const float *p = (const float *)&g_c;
    for (int i = 0; i < (int)(sizeof(wsm3_consts) / sizeof(float)); ++i) out[i] = p[i];
}

/* wsm3 (:74-216): t = th*pii, wsm32D per row, th = t/pii.  Arrays X(i,k,j) -> i + nx*(k + nz*j), 1-based inclusive tile bounds. */
int orc_wsm3(int nx, int nz, int ny, float *th, float *q, float *qci, float *qrs, const float *w, const float *den, const float *pii,
             const float *p, const float *delz, const float *args18, float *rain, float *rainncv, float *snow, float *snowncv, float *sr,
             int its, int ite, int jts, int jte, int kts, int kte)
{
    wsm3_args A;
    const int km = kte - kts + 1;
    if (km > W3_MAXK || km < 3) return 1;
    { float *a = (float *)&A; for (int i = 0; i < 18; ++i) a[i] = args18[i]; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = jts - 1; j <= jte - 1; ++j) for (int i = its - 1; i <= ite - 1; ++i) {
        float t[W3_MAXK], cq[W3_MAXK], cqci[W3_MAXK], cqrs[W3_MAXK], cw[W3_MAXK], cden[W3_MAXK], cp[W3_MAXK], cdz[W3_MAXK];
        for (int k = 0; k < km; ++k) {
            const size_t c = (size_t)i + (size_t)nx * ((size_t)(k + kts - 1) + (size_t)nz * j);
            t[k] = th[c] * pii[c]; cq[k] = q[c]; cqci[k] = qci[c]; cqrs[k] = qrs[c]; cw[k] = w[c]; cden[k] = den[c]; cp[k] = p[c]; cdz[k] = delz[c];
        }
        const size_t o = (size_t)i + (size_t)nx * j;
        wsm3_column(&g_c, &A, km, t, cq, cqci, cqrs, cw, cden, cp, cdz, &rain[o], &rainncv[o], &snow[o], &snowncv[o], &sr[o]);
        for (int k = 0; k < km; ++k) {
            const size_t c = (size_t)i + (size_t)nx * ((size_t)(k + kts - 1) + (size_t)nz * j);
            th[c] = t[k] / pii[c]; q[c] = cq[k]; qci[c] = cqci[k]; qrs[c] = cqrs[k];
        }
    }
    return 0;
}
