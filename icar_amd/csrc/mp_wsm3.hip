// icar_amd/csrc/mp_wsm3.hip -- WSM3 microphysics (src/physics/mp_wsm3.f90), SURVEY 8(f) rank 4, behind mp()'s dispatch
// (mp_driver.f90:552-585).  First device version: one column per thread, the column routine of wsm3_column.h with its
// per-level work arrays in private (scratch) memory -- correct first, not yet laid out like Thompson / mp_simple
// (one level per thread).  REAL(4) exp / log / x**y are the FP64 function rounded once (fp64_math.h), sqrt and divide IEEE:
// the oracle's math mode 1 evaluates the same column routine with the same definition of the transcendentals.
#include "ctx.h"
#include "fp64_math.h"
#include <cmath>

namespace {
__device__ __forceinline__ float w3_expf(float x) { return (float)d_exp((double)x); }
__device__ __forceinline__ float w3_logf(float x)
{
    if (x > 0.0f) return (float)d_log((double)x);
    return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
}
__device__ __forceinline__ float w3_powf(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x > 0.0f) return (float)d_exp((double)y * d_log((double)x));
    return x == 0.0f ? (y > 0.0f ? 0.0f : __builtin_inff()) : __builtin_nanf("");
}
}  // namespace

#define W3_FN __host__ __device__ static inline
#ifdef __HIP_DEVICE_COMPILE__
#define W3_EXP(x) w3_expf(x)
#define W3_LOG(x) w3_logf(x)
#define W3_POW(x, y) w3_powf(x, y)
#define W3_SQRT(x) sqrtf(x)
#else                                   /* the host pass only needs the file to parse: the column code never runs there */
#define W3_EXP(x) expf(x)
#define W3_LOG(x) logf(x)
#define W3_POW(x, y) powf(x, y)
#define W3_SQRT(x) sqrtf(x)
#endif
#define W3_MAXK 64
#define W3_HOST_INIT
#include "wsm3_column.h"

struct Wsm3State { wsm3_consts c; bool ready = false; };

namespace {
__global__ void __launch_bounds__(64)
k_wsm3(Dims d, wsm3_consts C, wsm3_args A, float *__restrict__ th, float *__restrict__ q, float *__restrict__ qci, float *__restrict__ qrs,
       const float *__restrict__ w, const float *__restrict__ den, const float *__restrict__ pii, const float *__restrict__ p,
       const float *__restrict__ delz, double *__restrict__ precip_acc, double *__restrict__ snow_acc, int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y;
    if (i > i1) return;
    float t[W3_MAXK], cq[W3_MAXK], cqci[W3_MAXK], cqrs[W3_MAXK], cw[W3_MAXK], cden[W3_MAXK], cp[W3_MAXK], cdz[W3_MAXK];
    for (int k = 0; k < km; ++k) {                               // wsm3 (:151-155): t = th * pii
        const int c = d.idx(i, k0 + k, j);
        t[k] = th[c] * pii[c]; cq[k] = q[c]; cqci[k] = qci[c]; cqrs[k] = qrs[c]; cw[k] = w[c]; cden[k] = den[c]; cp[k] = p[c]; cdz[k] = delz[c];
    }
    float rain = 0.f, rainncv = 0.f, snow = 0.f, snowncv = 0.f, sr = 0.f;   // process_subdomain: precipitation = 0; snowfall = 0
    wsm3_column(&C, &A, km, t, cq, cqci, cqrs, cw, cden, cp, cdz, &rain, &rainncv, &snow, &snowncv, &sr);
    for (int k = 0; k < km; ++k) {                               // :171-175: th = t / pii
        const int c = d.idx(i, k0 + k, j);
        th[c] = t[k] / pii[c]; q[c] = cq[k]; qci[c] = cqci[k]; qrs[c] = cqrs[k];
    }
    const int c2 = i + d.nx * j;                                 // mp_driver.f90:587-595: REAL(8) accumulators += REAL(4)
    precip_acc[c2] = precip_acc[c2] + rain;
    snow_acc[c2] = snow_acc[c2] + snow;
}
}  // namespace

void icar_wsm3_free(icar_hip_ctx *c) { delete c->wsm3; c->wsm3 = nullptr; }

int icar_wsm3_init_run(icar_hip_ctx *c)
{
    // wsm3init(rhoair0, rhowater, rhosnow, cliq, cpv) as mp_driver.f90:105 calls it (wrf_constants.f90:30-35, :65-67)
    if (!c->wsm3) c->wsm3 = new Wsm3State;
    wsm3_init_consts(&c->wsm3->c, 1.28f, 1000.f, 100.f, 4190.f, 4.f * 461.6f);
    c->wsm3->ready = true;
    return 0;
}

int icar_wsm3_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte)
{
    if (!c->wsm3 || !c->wsm3->ready) { icar_set_error("wsm3: call icar_hip_wsm3_init first"); return 1; }
    if (its < c->ims || ite > c->ime || jts < c->jms || jte > c->jme || kts < c->kms || kte > c->kme) { icar_set_error("wsm3: tile outside memory bounds"); return 1; }
    if (ite < its || jte < jts) return 0;
    const int km = kte - kts + 1;
    if (km < 3 || km > W3_MAXK) { icar_set_error("wsm3: 3..64 levels in this build"); return 1; }
    float *th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE), *q = icar_field_f(c, ICAR_F_WATER_VAPOR);
    float *qci = icar_field_f(c, ICAR_F_CLOUD_WATER), *qrs = icar_field_f(c, ICAR_F_RAIN);
    const float *w = icar_field_f(c, ICAR_F_W_REAL), *den = icar_field_f(c, ICAR_F_DENSITY), *pii = icar_field_f(c, ICAR_F_EXNER);
    const float *p = icar_field_f(c, ICAR_F_PRESSURE), *dz = icar_field_f(c, ICAR_F_DZ_MASS);
    double *pa = (double *)icar_field_f(c, ICAR_F_PRECIPITATION, false), *sa = (double *)icar_field_f(c, ICAR_F_SNOWFALL, false);
    if (!th || !q || !qci || !qrs || !w || !den || !pii || !p || !dz || !pa || !sa) return 1;
    // what mp_driver.f90:554-585 passes: gravity, cp, cpv, Rd, Rw, 273.15, EP1, EP2, epsilon, XLS, XLV, XLF, rhoair0, rhowater,
    // cliq, cice, psat (icar_constants.f90:391-420, wrf_constants.f90:10-67)
    wsm3_args A;
    A.delt = dt; A.g = 9.81f; A.cpd = 1012.0f; A.cpv = 4.f * 461.6f; A.rd = 287.058f; A.rv = 461.5f; A.t0c = 273.15f;
    A.ep1 = 461.5f / 287.058f - 1.f; A.ep2 = 287.058f / 461.5f; A.qmin = 1.e-15f; A.xls = 2.85e6f; A.xlv0 = 2.5e6f; A.xlf0 = 3.50e5f;
    A.den0 = 1.28f; A.denr = 1000.f; A.cliq = 4190.f; A.cice = 2106.f; A.psat = 610.78f;
    ScopedTimer tm(c, "mp");
    dim3 g((ite - its + 1 + 63) / 64, jte - jts + 1), b(64);
    hipLaunchKernelGGL(k_wsm3, g, b, 0, c->stream, c->d, c->wsm3->c, A, th, q, qci, qrs, w, den, pii, p, dz, pa, sa,
                       its - c->ims, ite - c->ims, jts - c->jms, kts - c->kms, km);
    HIPCHK(hipGetLastError());
    return 0;
}
