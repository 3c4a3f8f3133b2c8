// icar_amd/csrc/mp_wsm3.hip -- WSM3 microphysics (src/physics/mp_wsm3.f90), SURVEY 8(f) rank 4, behind mp()'s dispatch
// (mp_driver.f90:552-585).  The level-local pieces of wsm3_column.h (state of a minor step; rates, update, condensation -- all the
// scheme's exp/log/pow) run one thread per cell; the column piece (the two semi-Lagrangian falls, the melting level, the
// surface flux) one thread per column, between them.  REAL(4) exp / log / x**y are the FP64 function rounded once (fp64_math.h), sqrt and divide IEEE:
// the oracle's math mode 1 evaluates the same column routine with the same definition of the transcendentals.
#include "ctx.h"
#include "glibc_flt32.h"
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace {
// REAL(4) exp / log / x**y as the compiled reference evaluates them: the C library's expf / logf / powf (glibc_flt32.h)
__device__ __forceinline__ float w3_expf(float x) { return gf_expf(x); }
__device__ __forceinline__ float w3_logf(float x) { return gf_logf(x); }
__device__ __forceinline__ float w3_powf(float x, float y) { return gf_powf(x, y); }
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
#include "wsm3_column.h"

// work arrays of one call, all (nx, nz, ny) REAL(4): the level pieces run one thread per CELL (10 M threads at 512x512x40: the
// transcendental-heavy part of the scheme, ~25 exp/log/pow per level), the fall / melt / surface piece one thread per COLUMN
struct Wsm3State {
    wsm3_consts c; bool ready = false;
    float *t = nullptr, *cpm = nullptr, *xl = nullptr, *denfac = nullptr, *qs = nullptr, *rh = nullptr, *vt = nullptr, *denqrs = nullptr,
          *vti = nullptr, *denqci = nullptr, *rain = nullptr, *snow = nullptr, *delq = nullptr, *zi = nullptr;
    size_t n3 = 0;
};

namespace {
struct W3Work { float *t, *cpm, *xl, *denfac, *qs, *rh, *vt, *denqrs, *vti, *denqci, *rain, *snow, *zi; };

// zi(k+1) = zi(k) + dz(k) (mp_wsm3.f90:1291-1294), the reference's running sum, once per call and column
__global__ void __launch_bounds__(64)
k_wsm3_zi(Dims d, const float *__restrict__ delz, float *__restrict__ zi, int i0, int i1, int j0, int k0, int km)
{
    const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y;
    if (i > i1) return;
    float run = 0.0f;
    for (int k = 0; k < km; ++k) { const int c = d.idx(i, k0 + k, j); run = run + delz[c]; zi[c] = run; }
}

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
        if (k == k0) { const int c2 = i + d.nx * j; W.rain[c2] = 0.f; W.snow[c2] = 0.f; }      // process_subdomain: precipitation = 0, snowfall = 0
        t = th[c] * pii[c];
        float cpm, xl;
        wsm3_level_init(&C, &A, q[c], t, &qci_, &qrs_, &cpm, &xl);
        W.t[c] = t; W.cpm[c] = cpm; W.xl[c] = xl; qci[c] = qci_; qrs[c] = qrs_;
    } else t = W.t[c];
    float denfac, qs, rh, vt, denqrs, vti, denqci;
    wsm3_level_prep(&C, &A, &S, t, q[c], qci_, qrs_, den[c], p[c], &denfac, &qs, &rh, &vt, &denqrs, &vti, &denqci);
    W.denfac[c] = denfac; W.qs[c] = qs; W.rh[c] = rh; W.vt[c] = vt; W.denqrs[c] = denqrs; W.vti[c] = vti; W.denqci[c] = denqci;
}

// ---- the fall with one WAVE per (column, species): lane = level (cell quantities) / interface (wi, zi, za, dza, qa, qmi, qpi
// live on lanes 0..km).  nislfv_rain_plm (mp_wsm3.f90:1266-1505) is sequential in k only in four places, which stay
// sequential here so that every sum and every comparison sees the reference's operands:
//   zi            running sum of dz (a lane-serial loop of km adds)
//   wi limiter    k = km..1 uses the wi(k+1) it may just have changed: evaluated for all k at once with the unmodified values;
//                 only from the highest level that trips the limit downward is it re-run serially (rare)
//   kb / kt       "first kk >= previous-1 with zi <= za(kk)": za is strictly increasing (the limiter guarantees dza >= 0.95 dz),
//                 so the first kk is the count of arrival heights below zi, the same for every start the reference can have;
//                 where kt is not found the reference's stale kt is < kb and the level gets qn = 0 either way
//   sums          the kb+1..kt-1 partial sums and the surface flux are short loops in k order
// Needs km + 1 <= 64 lanes; all cross-lane reads happen with every lane active.
// one (column, species) on one wave: lane = level for dz, den, denfac, tk, wwl (terminal velocity), rql (den*q); zi = height of
// interface `lane` (0 at lane 0).  Returns the surface flux integral; *qn_out = the fallen den*q of this lane's level.
// nearest-neighbour lane reads as DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1: one VALU slot; __shfl_up / __shfl_down are
// ds_bpermute_b32 at 24 cycles per wave, profiles/micro/valubench.hip).  A lane without a source reads 0; none of those values is used.
__device__ __forceinline__ float w3_up(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float w3_dn(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true)); }

__device__ __forceinline__ float w3_fall_wave_column(const wsm3_consts *C, int km, int lane, float dz, float den, float denfac, float tk,
                                                     float wwl, float rql, float zi, int iter, float dt, float *qn_out)
{
    const bool cell = lane < km;
    float precip = 0.0f, qn = rql;                                   // an empty column keeps den*q as it is (cycle i_loop)
    if (__ballot(cell && rql > 0.0f) != 0ull) {                      // allold > 0: den*q >= 0, so the sum is positive iff one term is
        float ww = cell ? wwl : 0.0f, wi, za, dza, qa;
        for (int n = 1;; ++n) {
            const float wm1 = w3_up(ww), wm2 = w3_up(wm1), wp1 = w3_dn(ww);
            const float fa1 = 9.f / 16.f, fa2 = 1.f / 16.f;
            if (lane == 0) wi = ww;
            else if (lane == 1) wi = 0.5f * (ww + wm1);
            else if (lane <= km - 2) wi = fa1 * (ww + wm1) - fa2 * (wp1 + wm2);
            else if (lane == km - 1) wi = 0.5f * (ww + wm1);
            else wi = wm1;                                           // lane == km: wi(km+1) = ww(km)
            if (lane >= 1 && lane < km && ww == 0.0f) wi = wm1;      // terminate at the top of the rain shaft
            const float con1 = 0.05f;                                // limiter, k = km-1 .. 0
            const float wip1 = w3_dn(wi);
            const float dec = (wip1 - wi) * dt / dz;
            const unsigned long long bad = __ballot(cell && dec > con1);
            if (bad) {                                               // wave-uniform
                // Serial only where it has to be: level k must be re-evaluated when wi(k+1) has just been changed; when a
                // level is left alone, everything below it still sees the values the parallel evaluation saw, so the walk
                // jumps to the next level that tripped there.
                const float cdz = con1 * dz / dt;                    // per lane, the reference's con1*dz(k)/dt
                unsigned long long rem = bad;
                int k = 63 - __builtin_clzll(rem);
                while (k >= 0) {
                    const float wk1 = __shfl(wi, k + 1), wk = __shfl(wi, k), dzk = __shfl(dz, k), ck = __shfl(cdz, k);
                    const float decfl = (wk1 - wk) * dt / dzk;
                    rem &= (k == 0) ? 0ull : ((1ull << k) - 1ull);   // levels below k that tripped with the unmodified values
                    if (decfl > con1) {                              // uniform: all lanes hold the same broadcast operands
                        if (lane == k) wi = wk1 - ck;
                        k = k - 1;
                    } else k = rem ? 63 - __builtin_clzll(rem) : -1;
                }
            }
            za = zi - wi * dt;                                       // interfaces 0..km
            const float zap1 = w3_dn(za);
            dza = (lane < km) ? zap1 - za : zi - za;                 // dza(km+1) = zi(km+1) - za(km+1)
            qa = cell ? rql * dz / dza : 0.0f;                       // qa(km+1) = 0
            if (n <= iter) {                                         // wave-uniform
                float r1, r2, r3, r4;
                const float wa = wsm3_slope1(C, cell ? qa / den : 0.f, den, denfac, tk, &r1, &r2, &r3, &r4);
                ww = cell ? 0.5f * (wwl + wa) : 0.0f;
                continue;
            }
            break;
        }
        float qmi = qa, qpi = qa;                                    // piecewise-linear reconstruction
        {
            const float qap1 = w3_dn(qa), qam1 = w3_up(qa), dzap1 = w3_dn(dza), dzam1 = w3_up(dza);
            if (lane >= 1 && lane < km) {
                const float dip = (qap1 - qa) / (dzap1 + dza);
                const float dim = (qa - qam1) / (dzam1 + dza);
                if (!(dip * dim <= 0.0f)) {
                    qpi = qa + 0.5f * (dip + dim) * dza;
                    qmi = 2.0f * qa - qpi;
                    if (qpi < 0.0f || qmi < 0.0f) { qpi = qa; qmi = qa; }
                }
            }
        }
        // interpolation to the regular grid: the output cell of this lane is [zi(lane), zi(lane+1)]
        const float zlo = zi, zhi = w3_dn(zi);
        const float za_top = __shfl(za, km);
        // arrival heights below zlo among interfaces 1..km (nb) and below zhi among 0..km-1 (nt): za increases strictly, so each
        // count is the position of the first za >= z -- a 6-step binary search per lane instead of km+1 comparisons
        int lo1 = 0, hi1 = km + 1, lo2 = 0, hi2 = km;
        for (int step = 0; step < 6; ++step) {
            const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
            const float v1 = __shfl(za, m1 < 63 ? m1 : 63), v2 = __shfl(za, m2 < 63 ? m2 : 63);
            if (lo1 < hi1) { if (v1 < zlo) lo1 = m1 + 1; else hi1 = m1; }
            if (lo2 < hi2) { if (v2 < zhi) lo2 = m2 + 1; else hi2 = m2; }
        }
        const float za0 = __shfl(za, 0);
        const int nb = lo1 - (za0 < zlo ? 1 : 0), nt = lo2;
        const bool live = cell && !(zlo >= za_top);                  // not yet `exit intp`
        const int kb = live ? nb + 1 : 1;                            // 1-based first kk with zi(k) <= za(kk+1); <= km when live
        const bool found = live && nt < km;                          // first kk with zi(k+1) <= za(kk) exists
        const int kt = found ? nt : 0;                               // that kk, minus 1
        const int ib = kb - 1, it = (kt >= 1 ? kt : 1) - 1;
        const float za_b = __shfl(za, ib), dza_b = __shfl(dza, ib), qpi_b = __shfl(qpi, ib), qmi_b = __shfl(qmi, ib), qa_b = __shfl(qa, ib);
        const float za_t = __shfl(za, it), dza_t = __shfl(dza, it), qpi_t = __shfl(qpi, it), qmi_t = __shfl(qmi, it);
        const float tl = (zlo - za_b) / dza_b;
        const float tl2 = tl * tl;
        const float qqd_b = 0.5f * (qpi_b - qmi_b);
        const float qql = qqd_b * tl2 + qmi_b * tl;
        float zsum = (1.f - tl) * dza_b, qsum = (qa_b - qql) * dza_b;
        const int cnt = (found && kt > kb) ? kt - kb - 1 : 0;        // m = kb+1 .. kt-1
        int cmax = cnt;
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(cmax, o); cmax = v > cmax ? v : cmax; }
        for (int s2 = 1; s2 <= cmax; ++s2) {
            const int m = kb + s2 - 1 <= 63 ? kb + s2 - 1 : 63;      // 0-based index of m = kb + s2
            const float dm = __shfl(dza, m), qm = __shfl(qa, m);
            if (s2 <= cnt) { zsum = zsum + dm; qsum = qsum + qm * dm; }
        }
        qn = 0.0f;
        if (found && kt == kb) {
            const float th = (zhi - za_b) / dza_b;
            const float th2 = th * th;
            const float qqh = qqd_b * th2 + qmi_b * th;
            qn = (qqh - qql) / (th - tl);
        } else if (found && kt > kb) {
            const float th = (zhi - za_t) / dza_t;
            const float th2 = th * th;
            const float qqd = 0.5f * (qpi_t - qmi_t);
            const float dqh = qqd * th2 + qmi_t * th;
            zsum = zsum + th * dza_t;
            qsum = qsum + dqh * dza_t;
            qn = qsum / zsum;
        }
        // rain out, k ascending (wave-uniform loop on broadcast values)
        for (int k = 0; k < km; ++k) {
            const float zk = __shfl(za, k), zk1 = __shfl(za, k + 1), qk = __shfl(qa, k), dk = __shfl(dza, k);
            if (zk < 0.0f && zk1 < 0.0f) { precip = precip + qk * dk; continue; }
            else if (zk < 0.0f && zk1 >= 0.0f) { precip = precip + qk * (0.0f - zk); break; }
            break;
        }
    }
    *qn_out = qn;
    return precip;
}

// a block = 4 waves = one row segment of W3_TC columns of one species: the seven column arrays are staged through LDS as
// [level][column] tiles (coalesced 128-B row reads; the column-per-wave access pattern itself would touch one cache line per
// lane), each wave then walks its W3_TC/4 columns with lane = level, and the results go back the same way.
#define W3_TC 16      // (16 columns = 15 kB of LDS per block: more blocks per CU than with 32; measured with mp_wsm6.hip's falls)
__global__ void __launch_bounds__(256)
k_wsm3_fall_tile(Dims d, wsm3_consts C, W3Work W, float *__restrict__ qci, float *__restrict__ qrs, const float *__restrict__ den_,
                 const float *__restrict__ delz, float *__restrict__ delq, float dt, int i0, int i1, int j0, int k0, int km)
{
    extern __shared__ float w3_lds[];                                // [7][km][W3_TC + 1]
    const int ib = i0 + blockIdx.x * W3_TC, j = j0 + blockIdx.y;
    const bool ice = blockIdx.z == 1;
    const int ncol = min(W3_TC, i1 - ib + 1);
    const int LS = W3_TC + 1, plane = km * LS;
    float *__restrict__ qx = ice ? qci : qrs;
    float *__restrict__ denq = ice ? W.denqci : W.denqrs;
    const float *src[7] = {delz, den_, W.denfac, W.t, ice ? W.vti : W.vt, denq, W.zi};
    for (int a = 0; a < 7; ++a)
        for (int e = threadIdx.x; e < km * W3_TC; e += 256) {
            const int k = e / W3_TC, ci = e % W3_TC;
            if (ci < ncol) w3_lds[a * plane + k * LS + ci] = src[a][d.idx(ib + ci, k0 + k, j)];
        }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane < km ? lane : km - 1;                        // lanes beyond the column read a valid level (values unused)
    for (int t = 0; t < W3_TC / 4; ++t) {
        const int ci = wave * (W3_TC / 4) + t;
        if (ci >= ncol) break;                                       // wave-uniform
        const float dz = w3_lds[0 * plane + kl * LS + ci], den = w3_lds[1 * plane + kl * LS + ci], denfac = w3_lds[2 * plane + kl * LS + ci],
                    tk = w3_lds[3 * plane + kl * LS + ci], wwl = w3_lds[4 * plane + kl * LS + ci], rql = w3_lds[5 * plane + kl * LS + ci];
        const int kz = (lane <= km ? lane : km) - 1;
        const float zi = lane == 0 ? 0.0f : w3_lds[6 * plane + kz * LS + ci];
        float qn;
        const float precip = w3_fall_wave_column(&C, km, lane, dz, den, denfac, tk, wwl, rql, zi, ice ? 0 : 1, dt, &qn);
        if (lane < km) w3_lds[5 * plane + lane * LS + ci] = qn;      // this wave is the only reader / writer of column ci
        if (lane == 0) delq[(size_t)(ice ? 1 : 0) * d.nx * d.ny + (ib + ci) + d.nx * j] = precip;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < km * W3_TC; e += 256) {
        const int k = e / W3_TC, ci = e % W3_TC;
        if (ci < ncol) {
            const int c = d.idx(ib + ci, k0 + k, j);
            const float qn = w3_lds[5 * plane + k * LS + ci];
            denq[c] = qn;                                            // rql(i,:) = qn(:)
            qx[c] = w3_max(qn / w3_lds[1 * plane + k * LS + ci], 0.f);   // q = max(den*q / den, 0)
        }
    }
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
                    &c->wsm3->vti, &c->wsm3->denqci, &c->wsm3->rain, &c->wsm3->snow, &c->wsm3->delq, &c->wsm3->zi};
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
        float **p3[] = {&S->t, &S->cpm, &S->xl, &S->denfac, &S->qs, &S->rh, &S->vt, &S->denqrs, &S->vti, &S->denqci, &S->zi};
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
    W3Work W = {S->t, S->cpm, S->xl, S->denfac, S->qs, S->rh, S->vt, S->denqrs, S->vti, S->denqci, S->rain, S->snow, S->zi};
    ScopedTimer tm(c, "mp");
    // (the call's REAL(4) surface sums are zeroed per column by k_wsm3_prep: calls on disjoint tiles -- the strips and the interior on
    // the context's two streams -- share no scratch)
    const int i0 = its - c->ims, i1 = ite - c->ims, j0 = jts - c->jms, k0 = kts - c->kms, nxb = (ite - its + 1 + 63) / 64, nyt = jte - jts + 1;
    const dim3 gc(nxb, (km + 3) / 4, nyt), bc(64, 4), g2(nxb, nyt), b2(64);
    hipLaunchKernelGGL(k_wsm3_zi, g2, b2, 0, c->stream, c->d, dz, S->zi, i0, i1, j0, k0, km);
    for (int loop = 1; loop <= loops; ++loop) {
        if (loop == 1) hipLaunchKernelGGL((k_wsm3_prep<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, i0, i1, j0, k0, km);
        else           hipLaunchKernelGGL((k_wsm3_prep<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, i0, i1, j0, k0, km);
        if (km + 1 <= 64) {                                   // lane = level; more levels: one thread per column
            const int ncol_x = ite - its + 1;
            hipLaunchKernelGGL(k_wsm3_fall_tile, dim3((ncol_x + W3_TC - 1) / W3_TC, nyt, 2), dim3(256), 7 * (size_t)km * (W3_TC + 1) * sizeof(float),
                               c->stream, c->d, S->c, W, qci, qrs, den, dz, S->delq, dtcld, i0, i1, j0, k0, km);
        } else
            hipLaunchKernelGGL(k_wsm3_fall, dim3(nxb, nyt, 2), b2, 0, c->stream, c->d, S->c, W, qci, qrs, den, dz, S->delq, dtcld, i0, i1, j0, k0, km);
        hipLaunchKernelGGL(k_wsm3_melt, g2, b2, 0, c->stream, c->d, A, W, qci, qrs, w, den, dz, S->delq, dtcld, i0, i1, j0, k0, km);
        if (loop == loops) hipLaunchKernelGGL((k_wsm3_rates<true>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, dtcld, i0, i1, j0, k0, km);
        else               hipLaunchKernelGGL((k_wsm3_rates<false>), gc, bc, 0, c->stream, c->d, S->c, A, W, th, pii, q, qci, qrs, den, p, dtcld, i0, i1, j0, k0, km);
    }
    hipLaunchKernelGGL(k_wsm3_accumulate, g2, b2, 0, c->stream, c->d, W, pa, sa, i0, i1, j0);
    HIPCHK(hipGetLastError());
    return 0;
}
