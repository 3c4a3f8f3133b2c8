// icar_amd/csrc/mp_wsm3.hip -- WSM3 microphysics (src/physics/mp_wsm3.f90), SURVEY 8(f) rank 4, behind mp()'s dispatch
// (mp_driver.f90:552-585).  The level-local pieces of wsm3_column.h (state of a minor step; rates, update, condensation -- all the
// scheme's exp/log/pow) run one thread per cell; the column piece (the two semi-Lagrangian falls, the melting level, the
// surface flux) one thread per column, between them.  REAL(4) exp / log / x**y are the FP64 function rounded once (fp64_math.h), sqrt and divide IEEE:
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

// work arrays of one call, all (nx, nz, ny) REAL(4): the level pieces run one thread per CELL (10 M threads at 512x512x40: the
// transcendental-heavy part of the scheme, ~25 exp/log/pow per level), the fall / melt / surface piece one thread per COLUMN
struct Wsm3State {
    wsm3_consts c; bool ready = false;
    float *t = nullptr, *cpm = nullptr, *xl = nullptr, *denfac = nullptr, *qs = nullptr, *rh = nullptr, *vt = nullptr, *denqrs = nullptr,
          *vti = nullptr, *denqci = nullptr, *rain = nullptr, *snow = nullptr, *delq = nullptr;
    size_t n3 = 0;
};

namespace {
struct W3Work { float *t, *cpm, *xl, *denfac, *qs, *rh, *vt, *denqrs, *vti, *denqci, *rain, *snow; };

// per cell, top of a minor loop (first: also t = th*pii (:151-155), the clamps, cpm, xl)
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_wsm3_prep(Dims d, wsm3_consts C, wsm3_args A, W3Work W, const float *__restrict__ th, const float *__restrict__ pii, const float *__restrict__ q,
            float *__restrict__ qci, float *__restrict__ qrs, const float *__restrict__ den, const float *__restrict__ p,
            int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, k = k0 + blockIdx.y * 4 + threadIdx.y, j = j0 + blockIdx.z;
    if (i > i1 || k - k0 >= km) return;
    const int c = d.idx(i, k, j);
    const w3_sat S = wsm3_sat_coeffs(&A);
    float t, qci_ = qci[c], qrs_ = qrs[c];
    if (FIRST) {
        t = th[c] * pii[c];
        float cpm, xl;
        wsm3_level_init(&C, &A, q[c], t, &qci_, &qrs_, &cpm, &xl);
        W.t[c] = t; W.cpm[c] = cpm; W.xl[c] = xl; qci[c] = qci_; qrs[c] = qrs_;
    } else t = W.t[c];
    float denfac, qs, rh, vt, denqrs, vti, denqci;
    wsm3_level_prep(&C, &A, &S, t, q[c], qci_, qrs_, den[c], p[c], &denfac, &qs, &rh, &vt, &denqrs, &vti, &denqci);
    W.denfac[c] = denfac; W.qs[c] = qs; W.rh[c] = rh; W.vt[c] = vt; W.denqrs[c] = denqrs; W.vti[c] = vti; W.denqci[c] = denqci;
}

// per column and species (blockIdx.z: 0 rain/snow, 1 cloud ice): the semi-Lagrangian fall.  The column arrays are read and
// written in place (element stride nx, coalesced across the lanes of a wave); only the fall routine's nine work arrays are
// private.  The two species are independent until the melting level, which doubles the number of (serial) threads.
__global__ void __launch_bounds__(64)
k_wsm3_fall(Dims d, wsm3_consts C, W3Work W, float *__restrict__ qci, float *__restrict__ qrs, const float *__restrict__ den,
            const float *__restrict__ delz, float *__restrict__ delq, float dtcld, int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y;
    if (i > i1) return;
    const int c0 = d.idx(i, k0, j), c2 = i + d.nx * j;
    const bool ice = blockIdx.z == 1;
    const float r = wsm3_fall_species(&C, km, d.sk, dtcld, W.t + c0, (ice ? qci : qrs) + c0, den + c0, delz + c0, W.denfac + c0,
                                      (ice ? W.vti : W.vt) + c0, (ice ? W.denqci : W.denqrs) + c0, ice ? 0 : 1);
    delq[(size_t)blockIdx.z * d.nx * d.ny + c2] = r;
}

// per column: melting level and surface flux (this call's REAL(4) sums in W.rain / W.snow)
__global__ void __launch_bounds__(64)
k_wsm3_melt(Dims d, wsm3_args A, W3Work W, const float *__restrict__ qci, const float *__restrict__ qrs, const float *__restrict__ w,
            const float *__restrict__ den, const float *__restrict__ delz, const float *__restrict__ delq, float dtcld,
            int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y;
    if (i > i1) return;
    const int c0 = d.idx(i, k0, j), c2 = i + d.nx * j;
    float rain = W.rain[c2], snow = W.snow[c2], rainncv = 0.f, snowncv = 0.f, sr = 0.f;   // rainncv / snowncv / sr only feed sr, which ICAR drops
    wsm3_melt_surface(&A, km, d.sk, dtcld, delq[c2], delq[(size_t)d.nx * d.ny + c2], W.t + c0, qci + c0, qrs + c0, w + c0, den + c0, delz + c0,
                      W.cpm + c0, W.vt + c0, W.denqrs + c0, &rain, &rainncv, &snow, &snowncv, &sr);
    W.rain[c2] = rain; W.snow[c2] = snow;
}

// per cell: rates, update, condensation; LAST: th = t / pii (:171-175)
template <bool LAST>
__global__ void __launch_bounds__(256)
k_wsm3_rates(Dims d, wsm3_consts C, wsm3_args A, W3Work W, float *__restrict__ th, const float *__restrict__ pii, float *__restrict__ q,
             float *__restrict__ qci, float *__restrict__ qrs, const float *__restrict__ den, const float *__restrict__ p, float dtcld,
             int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, k = k0 + blockIdx.y * 4 + threadIdx.y, j = j0 + blockIdx.z;
    if (i > i1 || k - k0 >= km) return;
    const int c = d.idx(i, k, j);
    const w3_sat S = wsm3_sat_coeffs(&A);
    float t = W.t[c], q_ = q[c], qci_ = qci[c], qrs_ = qrs[c];
    wsm3_level_rates(&C, &A, &S, dtcld, &t, &q_, &qci_, &qrs_, den[c], p[c], W.denfac[c], W.qs[c], W.rh[c], W.cpm[c], W.xl[c]);
    q[c] = q_; qci[c] = qci_; qrs[c] = qrs_;
    if (LAST) th[c] = t / pii[c]; else W.t[c] = t;
}

// mp_driver.f90:587-595: REAL(8) accumulators += this call's REAL(4) precipitation / snowfall
__global__ void k_wsm3_accumulate(Dims d, W3Work W, double *__restrict__ precip_acc, double *__restrict__ snow_acc, int i0, int i1, int j0)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y;
    if (i > i1) return;
    const int c2 = i + d.nx * j;
    precip_acc[c2] = precip_acc[c2] + W.rain[c2];
    snow_acc[c2] = snow_acc[c2] + W.snow[c2];
}
}  // namespace

void icar_wsm3_free(icar_hip_ctx *c)
{
    if (!c->wsm3) return;
    float **ps[] = {&c->wsm3->t, &c->wsm3->cpm, &c->wsm3->xl, &c->wsm3->denfac, &c->wsm3->qs, &c->wsm3->rh, &c->wsm3->vt, &c->wsm3->denqrs,
                    &c->wsm3->vti, &c->wsm3->denqci, &c->wsm3->rain, &c->wsm3->snow, &c->wsm3->delq};
    for (float **p : ps) if (*p) hipFree(*p);
    delete c->wsm3; c->wsm3 = nullptr;
}

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
    Wsm3State *S = c->wsm3;
    if (!S->t) {
        float **p3[] = {&S->t, &S->cpm, &S->xl, &S->denfac, &S->qs, &S->rh, &S->vt, &S->denqrs, &S->vti, &S->denqci};
        for (float **x : p3) HIPCHK(hipMalloc(x, c->n3 * sizeof(float)));
        HIPCHK(hipMalloc(&S->rain, (size_t)c->d.nx * c->d.ny * sizeof(float))); HIPCHK(hipMalloc(&S->snow, (size_t)c->d.nx * c->d.ny * sizeof(float)));
        HIPCHK(hipMalloc(&S->delq, 2 * (size_t)c->d.nx * c->d.ny * sizeof(float)));
    }
    // what mp_driver.f90:554-585 passes: gravity, cp, cpv, Rd, Rw, 273.15, EP1, EP2, epsilon, XLS, XLV, XLF, rhoair0, rhowater,
    // cliq, cice, psat (icar_constants.f90:391-420, wrf_constants.f90:10-67)
    wsm3_args A;
    A.delt = dt; A.g = 9.81f; A.cpd = 1012.0f; A.cpv = 4.f * 461.6f; A.rd = 287.058f; A.rv = 461.5f; A.t0c = 273.15f;
    A.ep1 = 461.5f / 287.058f - 1.f; A.ep2 = 287.058f / 461.5f; A.qmin = 1.e-15f; A.xls = 2.85e6f; A.xlv0 = 2.5e6f; A.xlf0 = 3.50e5f;
    A.den0 = 1.28f; A.denr = 1000.f; A.cliq = 4190.f; A.cice = 2106.f; A.psat = 610.78f;
    int loops; const float dtcld = wsm3_dtcld(&A, &loops);
    W3Work W = {S->t, S->cpm, S->xl, S->denfac, S->qs, S->rh, S->vt, S->denqrs, S->vti, S->denqci, S->rain, S->snow};
    ScopedTimer tm(c, "mp");
    HIPCHK(hipMemsetAsync(S->rain, 0, (size_t)c->d.nx * c->d.ny * sizeof(float), c->stream));     // process_subdomain: precipitation = 0
    HIPCHK(hipMemsetAsync(S->snow, 0, (size_t)c->d.nx * c->d.ny * sizeof(float), c->stream));
    const int i0 = its - c->ims, i1 = ite - c->ims, j0 = jts - c->jms, k0 = kts - c->kms, nxb = (ite - its + 1 + 63) / 64, nyt = jte - jts + 1;
    const dim3 gc(nxb, (km + 3) / 4, nyt), bc(64, 4), g2(nxb, nyt), b2(64);
    for (int loop = 1; loop <= loops; ++loop) {
        if (loop == 1) hipLaunchKernelGGL((k_wsm3_prep<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, i0, i1, j0, k0, km);
        else           hipLaunchKernelGGL((k_wsm3_prep<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, i0, i1, j0, k0, km);
        hipLaunchKernelGGL(k_wsm3_fall, dim3(nxb, nyt, 2), b2, 0, c->stream, c->d, S->c, W, qci, qrs, den, dz, S->delq, dtcld, i0, i1, j0, k0, km);
        hipLaunchKernelGGL(k_wsm3_melt, g2, b2, 0, c->stream, c->d, A, W, qci, qrs, w, den, dz, S->delq, dtcld, i0, i1, j0, k0, km);
        if (loop == loops) hipLaunchKernelGGL((k_wsm3_rates<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, dtcld, i0, i1, j0, k0, km);
        else               hipLaunchKernelGGL((k_wsm3_rates<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, dtcld, i0, i1, j0, k0, km);
    }
    hipLaunchKernelGGL(k_wsm3_accumulate, g2, b2, 0, c->stream, c->d, W, pa, sa, i0, i1, j0);
    HIPCHK(hipGetLastError());
    return 0;
}
