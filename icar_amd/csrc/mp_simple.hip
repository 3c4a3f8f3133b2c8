// icar_amd/csrc/mp_simple.hip -- ICAR's "simple" (SB04) bulk microphysics on gfx950 (row M1).
//
// Reference algorithm: src/physics/mp_simple.f90 (driver :595-646, column :481-566, per-level
// conversions :381-420, saturation adjustment :198-280, sedimentation :437-459).
// Design (k_mp_simple_pack): one LEVEL per thread, whole columns packed into a 256-thread block (thread = level*cpb +
// column, column_comm.h).  The scheme is per-level work -- the saturation iteration (:198-280), the conversions and the
// evaporation sweep after every fall sub-step -- coupled only by the fall fluxes: flux(k+1) of the level above is read
// through LDS once per sub-step (mp_simple.f90:437-459 only ever uses the not-yet-modified q(k+1), so all levels of a
// sub-step are independent), and the column-wide "any rain / snow" and CFL numbers are LDS reductions.  State lives in
// registers; 16 waves per CU instead of the 3 a one-column-per-lane layout (51 kB of LDS per 64 columns) could keep
// resident.  FP32 throughout like the reference;
// exp() is evaluated in FP64 and rounded once so that it agrees with the host libm's correctly
// rounded expf in all but ~1e-3 of evaluations (documented tolerance in tests/).
#include "ctx.h"
#include "glibc_flt32.h"
#include "column_comm.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {
constexpr float LH_vapor = 2.26E6f, dLHvdt = 2400.0f, LH_liquid = 3.34E5f, heat_capacity = 1006.0f;
constexpr float SMALL_VALUE = 1E-30f, freezing_threshold = 273.15f;
constexpr float snow_fall_rate = 1.5f, rain_fall_rate = 10.0f, snow_cloud_init = 0.0001f, rain_cloud_init = 0.0001f;

__device__ __forceinline__ float expf_cr(float x) { return gf_expf(x); }          // the C library's expf, bit for bit (glibc_flt32.h)

__device__ __forceinline__ float sat_mr(float temperature, float pressure)
{   // mp_simple.f90:146-182
    float a, b;
    if (temperature < freezing_threshold) { a = 21.8745584f; b = 7.66f; }
    else { a = 17.2693882f; b = 35.86f; }
    float e_s = 610.78f * expf_cr(a * (temperature - 273.16f) / (temperature - b));
    if ((pressure - e_s) <= 0) e_s = pressure * 0.99999f;
    return 0.6219907f * e_s / (pressure - e_s);
}

__device__ __forceinline__ void phase_change(float &temperature, float &q1, float qmax, float &q2,
                                             float Lheat, float change_rate, int &err)
{   // :333-362
    const float mass2temp = Lheat / heat_capacity;
    float delta = (qmax - q2) * change_rate;
    if (delta > q1) delta = q1;
    if (delta > ((qmax - q2) * 0.99f)) delta = (qmax - q2) * 0.99f;
    q1 = q1 - delta;
    if (q1 < 0) {
        if ((q1 + SMALL_VALUE) < 0) q1 = 0;
        else err = 1;                       // the reference prints and STOPs here
    }
    q2 = q2 + delta;
    temperature = temperature + delta * mass2temp;
}

__device__ __forceinline__ void cloud2hydrometeor(float &qc, float &q, float conversion, float qcmin)
{   // :295-315
    float delta = (qc > qcmin) ? qc - (qc * conversion) : 0.0f;
    if (delta < qc) { qc = qc - delta; q = q + delta; }
    else { q = q + qc; qc = 0.f; }
    qc = fmaxf(qc, 0.f);
}

__device__ void mp_conversions(float pressure, float &temperature, float &qv, float &qc, float &qr, float &qs,
                               float cloud2rain, float cloud2snow, int &err)
{   // :381-420 with cloud_conversion (:198-280) inlined
    const float L_melt = -1 * LH_liquid;
    const float L_evap = -1 * (LH_vapor + (373.15f - temperature) * dLHvdt);
    const float L_subl = L_melt + L_evap;
    float qvsat = 0.0f;
    {
        const float maxerr = 1e-4f;
        int iteration = 0;
        float lastqv = qv + maxerr * 2;
        const float vapor2temp = (LH_vapor + (373.15f - temperature) * dLHvdt) / heat_capacity;
        const float pre_qc = qc, pre_t = temperature;
        while ((fabsf(lastqv - qv) > maxerr) && (iteration < 15)) {
            iteration = iteration + 1;
            lastqv = qv;
            qvsat = sat_mr(temperature, pressure);
            if (qv > qvsat) {
                const float excess = (qv - qvsat) * 0.5f;
                temperature = temperature + (excess * vapor2temp);
                qv = qv - excess;
                qc = qc + excess;
            } else if (qc > 0) {
                const float excess = (qvsat - qv) * 0.5f;
                if (excess < qc) {
                    temperature = temperature - (excess * vapor2temp);
                    qv = qv + excess;
                    qc = qc - excess;
                } else {
                    qv = qv + qc;
                    temperature = temperature - (qc * vapor2temp);
                    qc = 0.f;
                }
            }
        }
        if (iteration == 15) {
            qv = sat_mr(pre_t, pressure);
            temperature = pre_t;
            qc = pre_qc;
        }
        qc = fmaxf(qc, 0.f);
    }
    if ((qc + qr + qs) > SMALL_VALUE) {
        if (qc > SMALL_VALUE) {
            if (temperature > freezing_threshold) {
                cloud2hydrometeor(qc, qr, cloud2rain, rain_cloud_init);
                if (qs > SMALL_VALUE) phase_change(temperature, qs, 100.f, qr, L_melt, cloud2rain, err);
            } else
                cloud2hydrometeor(qc, qs, cloud2snow, snow_cloud_init);
        }
        if (qv < qvsat) {
            if (qr > SMALL_VALUE) phase_change(temperature, qr, qvsat, qv, L_evap, cloud2rain / 2, err);
            if (qs > SMALL_VALUE) phase_change(temperature, qs, qvsat, qv, L_subl, cloud2snow / 2, err);
        }
    }
}

// one level of one column per thread; see the header comment.  All threads of the block run the same sub-step loops.
// up to 4 tiles (the strips of process_halo, mp_driver.f90:609-658) in one launch: blockIdx.z = tile
struct MpTiles { int i0[4], i1[4], j0[4], nrow[4], ib0[4], nb[4]; };

__global__ void __launch_bounds__(1024, 4)
k_mp_simple_pack(Dims d, const float *__restrict__ pressure, float *__restrict__ th, const float *__restrict__ pii,
                 const float *__restrict__ rho, float *__restrict__ qv_g, float *__restrict__ qc_g,
                 float *__restrict__ qr_g, float *__restrict__ qs_g, const float *__restrict__ dz,
                 double *__restrict__ precip_acc, double *__restrict__ snow_acc,
                 float dt, float cloud2rain, float cloud2snow,
                 MpTiles tl, int kts, int kte, int cpb, int *__restrict__ err_count)
{
    extern __shared__ double lds_pack[];
    const int nz = d.nz;
    const int z = blockIdx.z;                                     // strip of process_halo (or the one tile)
    if ((int)blockIdx.x >= tl.nb[z] || (int)blockIdx.y >= tl.nrow[z]) return;
    const int i0 = tl.i0[z], i1 = tl.i1[z];
    const int first = (tl.ib0[z] + blockIdx.x) * cpb;
    BlockComm x(lds_pack, threadIdx.x, blockDim.x, cpb, nz, i0 - first, i1 - first);
    const int j = tl.j0[z] + blockIdx.y;
    const int i = x.active ? first + x.col : max(i0, min(i1, first));
    const int k = x.k;
    const int c = d.idx(i, k, j);
    const float pi_ = pii[c], p = pressure[c], rh = rho[c], dzk = dz[c];
    float T = th[c] * pi_, qv = qv_g[c], qc = qc_g[c], qr = qr_g[c], qs = qs_g[c];
    int err = 0;
    float rain = 0.0f, snow = 0.0f;
    const float L_melt = -1 * LH_liquid;
    const bool in_k = x.active && k >= kts && k <= kte;
    const int top = (kte < nz - 2) ? kte : nz - 2;
    if (in_k) mp_conversions(p, T, qv, qc, qr, qs, cloud2rain, cloud2snow, err);
    // saturation mixing ratio of the fall sub-steps (:516-528, :543-562): a level whose temperature did not change since the last
    // evaluation (saturated air: no evaporation) would compute the same exp again, 14 + 3 times at dt = 68 s
    float T_sat = -1.0f, qvsat_c = 0.0f;
    // one fall sub-step of species q (:437-459): F = flux leaving this level, upF = flux arriving from the level above
#define MPS_SEDIMENT(q, vfall, acc)                                                                   \
    {                                                                                                 \
        const float F = vfall * q * rh;                                                               \
        const float upF = x.up1(F);                                                                   \
        if (on && x.active) {                                                                         \
            if (k == kts) { q = q - (F / dzk / rh); acc; }                                            \
            else if (k > kts && k <= top + 1) q = q - F / (rh * dzk);                                 \
            if (k >= kts && k <= top) q = q + upF / (rh * dzk);                                       \
        }                                                                                             \
    }
    {   // rain :503-530
        const bool has = x.col_any(x.active && qr > SMALL_VALUE);
        const float m = x.col_max_pos(x.active ? dt / dzk * rain_fall_rate : 0.0f);
        const float cfl = ceilf(m);
        const float vfall = dt * rain_fall_rate / cfl;
        const int ncfl = has ? (int)lroundf(cfl) : 0;
        const float rate = cloud2rain / (2 * ncfl);
        for (int nmax = x.loop_max(ncfl), s = 1; s <= nmax; ++s) {
            const bool on = s <= ncfl;
            MPS_SEDIMENT(qr, vfall, rain = rain + F)
            if (on && in_k && qr > SMALL_VALUE) {             // (the reference evaluates sat_mr first; its value is only read here)
                const float L_evap = -1 * (LH_vapor + (373.15f - T) * dLHvdt);
                if (T != T_sat) { qvsat_c = sat_mr(T, p); T_sat = T; }      // same T, same p: the same number as a new evaluation
                if (qv < qvsat_c) phase_change(T, qr, qvsat_c, qv, L_evap, rate, err);
            }
        }
    }
    {   // snow :531-562
        const bool has = x.col_any(x.active && qs > SMALL_VALUE);
        const float m = x.col_max_pos(x.active ? dt / dzk * snow_fall_rate : 0.0f);
        const float cfl = ceilf(m);
        const float vfall = dt * snow_fall_rate / cfl;
        const int ncfl = has ? (int)lroundf(cfl) : 0;
        const float rate = cloud2snow / (2 * ncfl);
        for (int nmax = x.loop_max(ncfl), s = 1; s <= nmax; ++s) {
            const bool on = s <= ncfl;
            MPS_SEDIMENT(qs, vfall, { snow = snow + F; rain = rain + F; })
            if (on && in_k && qs > SMALL_VALUE) {
                const float L_evap = -1 * (LH_vapor + (373.15f - T) * dLHvdt);
                const float L_subl = L_melt + L_evap;
                if (T != T_sat) { qvsat_c = sat_mr(T, p); T_sat = T; }
                if (qv < qvsat_c) phase_change(T, qs, qvsat_c, qv, L_subl, rate, err);
            }
        }
    }
#undef MPS_SEDIMENT
    if (!x.active) return;
    th[c] = T / pi_;
    qv_g[c] = qv; qc_g[c] = qc; qr_g[c] = qr; qs_g[c] = qs;
    if (k == kts) {   // process_subdomain, mp_driver.f90:587-595: REAL(8) accumulators += REAL(4) tile fluxes
        const int c2 = i + d.nx * j;
        precip_acc[c2] = precip_acc[c2] + rain;
        snow_acc[c2] = snow_acc[c2] + snow;
    }
    if (err && err_count) atomicAdd(err_count, 1);
}
}  // namespace

int icar_mp_simple_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte, int *err_out)
{
    if (ntiles < 0 || ntiles > 4) { icar_set_error("mp_simple: 0..4 tiles per call"); return 1; }
    if (kts < c->kms || kte > c->kme) { icar_set_error("mp_simple: tile outside memory bounds"); return 1; }
    if (err_out) *err_out = 0;
    const int nz = c->d.nz;
    int nt = 0, cpb = 0;
    if (!(block_comm_geometry(nz, nt, cpb) > 0.0f)) { icar_set_error("mp_simple: more than 1024 levels are not supported"); return 1; }
    // whole columns packed into blocks of 256 threads (512 / 1024 when nz needs it), thread = level*cpb + column
    MpTiles tl; int n = 0, nbmax = 0, nrmax = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int its = tiles[t][0], ite = tiles[t][1], jts = tiles[t][2], jte = tiles[t][3];
        if (its < c->ims || ite > c->ime || jts < c->jms || jte > c->jme) { icar_set_error("mp_simple: tile outside memory bounds"); return 1; }
        if (ite < its || jte < jts) continue;
        for (int o = 0; o < n; ++o)                                // columns are updated in place: two tiles on one column would race
            if (its - c->ims <= tl.i1[o] && ite - c->ims >= tl.i0[o] && jts - c->jms < tl.j0[o] + tl.nrow[o] && jte - c->jms >= tl.j0[o]) {
                icar_set_error("mp_simple: tiles of one call must not overlap"); return 1;
            }
        tl.i0[n] = its - c->ims; tl.i1[n] = ite - c->ims; tl.j0[n] = jts - c->jms; tl.nrow[n] = jte - jts + 1;
        tl.ib0[n] = tl.i0[n] / cpb; tl.nb[n] = tl.i1[n] / cpb - tl.ib0[n] + 1;
        nbmax = std::max(nbmax, tl.nb[n]); nrmax = std::max(nrmax, tl.nrow[n]); ++n;
    }
    if (n == 0 || kte < kts) return 0;
    float *p = icar_field_f(c, ICAR_F_PRESSURE), *th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE);
    float *pii = icar_field_f(c, ICAR_F_EXNER), *rho = icar_field_f(c, ICAR_F_DENSITY);
    float *qv = icar_field_f(c, ICAR_F_WATER_VAPOR), *qc = icar_field_f(c, ICAR_F_CLOUD_WATER);
    float *qr = icar_field_f(c, ICAR_F_RAIN), *qs = icar_field_f(c, ICAR_F_SNOW), *dz = icar_field_f(c, ICAR_F_DZ_MASS);
    double *pa = (double *)icar_field_f(c, ICAR_F_PRECIPITATION, false), *sa = (double *)icar_field_f(c, ICAR_F_SNOWFALL, false);
    if (!p || !th || !pii || !rho || !qv || !qc || !qr || !qs || !dz || !pa || !sa) return 1;
    // mp_simple.f90:619-620, evaluated with the host libm like the reference
    const float cloud2snow = std::exp(-1.0f * (1 / 2000.0f) * dt);
    const float cloud2rain = std::exp(-1.0f * (1 / 500.0f) * dt);
    if (err_out) HIPCHK(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    ScopedTimer t(c, "mp");
    hipLaunchKernelGGL(k_mp_simple_pack, dim3(nbmax, nrmax, n), dim3(nt), BlockComm::lds_bytes(nt, cpb), c->stream, c->d,
                       p, th, pii, rho, qv, qc, qr, qs, dz, pa, sa, dt, cloud2rain, cloud2snow,
                       tl, kts - c->kms, kte - c->kms, cpb, err_out ? c->d_flag : nullptr);
    HIPCHK(hipGetLastError());
    if (err_out) {
        HIPCHK(hipMemcpyAsync(err_out, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int icar_mp_simple_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte, int *err_out)
{
    const int tile[1][4] = {{its, ite, jts, jte}};
    return icar_mp_simple_run_tiles(c, dt, 1, tile, kts, kte, err_out);
}
