// icar_amd/csrc/step.hip -- the streaming kernels that sit between the hot kernels inside step():
// rows T3 (diagnostic_update, src/main/time_step.f90:49-198) and F1 (apply_forcing / enforce_limits,
// src/objects/domain_obj.f90:2383-2448, 2228-2243).  All HBM-bound, lanes along i.
#include "ctx.h"
#include "glibc_flt32.h"
#include <cmath>

namespace {
constexpr float Rd = 287.058f, cp = 1012.0f;     // src/constants/icar_constants.f90:391-393

// (p/po)**(Rd/cp): the C library's powf, bit for bit what the compiled reference computes (glibc_flt32.h)
__device__ __forceinline__ float exner_function(float pressure)
{   // atm_utilities.f90:682-691 ; po = 100000 (integer in the reference => p/100000.)
    return gf_powf(pressure / 100000.0f, Rd / cp);
}

// diagnostic_update's thermodynamics in two launches:
//   k_diag_cell  exner, temperature, density of every cell (:87-101): one pow per cell, 2 arrays in, 3 out -- what the
//                microphysics that follows reads (exner) or is read from before the microphysics moves it (th);
//   k_diag_face  interface pressure / temperature, surface pressure, mass-point winds (:88-108): neighbours of p and of the
//                T just written, u, v -- nothing the microphysics touches, so the sub-step issues it on the second stream beside
//                the interior launch (timestep.hip).
// T(k+-1) read back from memory is the same th * exner the cell kernel stored, so the split changes no result.
__global__ void __launch_bounds__(256)
k_diag_cell(size_t n4, size_t n, const float *__restrict__ p, const float *__restrict__ th, float *__restrict__ exner,
            float *__restrict__ T, float *__restrict__ rho)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n4) {
        const float4 pc = ((const float4 *)p)[t], tc = ((const float4 *)th)[t];
        float4 ex, tt, rr;
        ex.x = exner_function(pc.x); ex.y = exner_function(pc.y); ex.z = exner_function(pc.z); ex.w = exner_function(pc.w);
        tt.x = tc.x * ex.x; tt.y = tc.y * ex.y; tt.z = tc.z * ex.z; tt.w = tc.w * ex.w;                     // :95
        rr.x = pc.x / (Rd * tt.x); rr.y = pc.y / (Rd * tt.y); rr.z = pc.z / (Rd * tt.z); rr.w = pc.w / (Rd * tt.w);   // :101
        ((float4 *)exner)[t] = ex; ((float4 *)T)[t] = tt; ((float4 *)rho)[t] = rr;
    } else {
        const size_t c = 4 * n4 + (t - n4);                       // the cells left over when n is not a multiple of 4
        if (c < n) {
            const float pc = p[c], ex = exner_function(pc), tt = th[c] * ex;
            exner[c] = ex; T[c] = tt; rho[c] = pc / (Rd * tt);
        }
    }
}

__global__ void __launch_bounds__(256)
k_diag_face(Dims d, const float *__restrict__ p, const float *__restrict__ T, float *__restrict__ p_i, float *__restrict__ psfc,
            float *__restrict__ T_i, const float *__restrict__ u, const float *__restrict__ v,
            float *__restrict__ u_mass, float *__restrict__ v_mass)
{
    const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y, j = blockIdx.z;
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    // interface values (:88-98): level kms extrapolates from kms+1, the others average with the level below
    const float pc = p[c], t = T[c];
    float pn, tn;
    if (k == 0) {
        const float p1 = p[c + d.sk], t1 = T[c + d.sk];
        pn = pc + (pc - p1) / 2; tn = t + (t - t1) / 2;
        psfc[i + d.nx * j] = pn;
    } else {
        const float pm = p[c - d.sk], tm = T[c - d.sk];
        pn = (pm + pc) / 2; tn = (tm + t) / 2;
    }
    p_i[c] = pn;
    T_i[c] = tn;
    if (u_mass) { const int cu = i + (d.nx + 1) * (k + d.nz * j); u_mass[c] = (u[cu + 1] + u[cu]) / 2; }   // :105
    if (v_mass) v_mass[c] = (v[c + d.sj] + v[c]) / 2;                                                       // :108
}

// w_real (:165-194): real vertical motion on interior cells, k-sequential through lastw.
// At most 16 VGPRs (the attribute counts half of gfx90a+'s unified file) and no unrolling: the host issues this kernel beside the
// MPDATA launch, whose persistent blocks leave exactly one more wave of 16 registers per SIMD (mpdata.hip).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(8)))
k_diag_wreal(Dims d, const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w,
             const float *__restrict__ dzdx, const float *__restrict__ dzdy, const float *__restrict__ jaco,
             float *__restrict__ w_real)
{
    const int i = 1 + blockIdx.x * 64 + threadIdx.x, j = 1 + blockIdx.y;
    if (i >= d.nx - 1) return;
    // raw buffer accesses: descriptor + per-lane byte offset (the column) + scalar byte offset (row and level), so that an
    // address costs one VGPR, not two per array (fields of 2 GiB or more are refused by the host)
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto mk = [](const float *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, -1, 0x00020000); };
    const rsrc_t ru = mk(u), rv = mk(v), rw = mk(w), rdx = mk(dzdx), rdy = mk(dzdy), rj = mk(jaco), ro = mk(w_real);
    auto ld = [](rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
    const int vi = 4 * i;
    const int nxu = d.nx + 1;
    float lastw = 0.0f;
#pragma unroll 1
    for (int k = 0; k < d.nz; ++k) {
        const int sc = 4 * d.idx(0, k, j), scu = 4 * (nxu * (k + d.nz * j));     // wave-uniform
        const float uw0 = ld(ru, vi, scu) * ld(rdx, vi, scu), uw1 = ld(ru, vi, scu + 4) * ld(rdx, vi, scu + 4);   // uw(i), uw(i+1)
        const float vw0 = ld(rv, vi, sc) * ld(rdy, vi, sc), vw1 = ld(rv, vi, sc + 4 * d.sj) * ld(rdy, vi, sc + 4 * d.sj);   // vw(j), vw(j+1)
        const float currw = ld(rw, vi, sc);
        const float r = (uw0 + uw1) * 0.5f + (vw0 + vw1) * 0.5f + ld(rj, vi, sc) * (lastw + currw) * 0.5f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, r), ro, vi, sc, 0);
        lastw = currw;
    }
}

// optional column integrals (:126-144 -> compute_ivt / compute_iq, atm_utilities.f90:35-102): one thread per column,
// levels kms..kme-1 bottom-up in the reference's accumulation order; the layer is cut at 500 hPa
struct ColumnArgs { const float *qv, *um, *vm, *p_i, *liq[2], *ice[3]; float *ivt, *iwv, *iwl, *iwi; };

__global__ void __launch_bounds__(64)
k_diag_columns(Dims d, ColumnArgs a)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
    if (i >= d.nx) return;
    const float gravity = 9.81f;                                    // icar_constants.f90
    float ivt = 0.0f, iwv = 0.0f, iwl = 0.0f, iwi = 0.0f;
    float pk = a.p_i[d.idx(i, 0, j)];
    for (int k = 0; k < d.nz - 1; ++k) {
        const int c = d.idx(i, k, j);
        const float pk1 = a.p_i[c + d.sk];
        float dp; bool use = true;
        if (pk1 > 50000.0f) dp = pk - pk1;
        else if (pk > 50000.0f) dp = pk - 50000.0f;
        else { dp = 0.0f; use = false; }
        if (use) {
            if (a.ivt) { const float u = a.um[c], v = a.vm[c]; ivt = ivt + (a.qv[c] * sqrtf(u * u + v * v) * dp) / gravity; }
            if (a.iwv) iwv = iwv + (a.qv[c] * dp) / gravity;
            if (a.iwl) {
                float t = 0.0f;
                if (a.liq[0]) t = t + a.liq[0][c];
                if (a.liq[1]) t = t + a.liq[1][c];
                iwl = iwl + (t * dp) / gravity;
            }
            if (a.iwi) {
                float t = 0.0f;
                if (a.ice[0]) t = t + a.ice[0][c];
                if (a.ice[1]) t = t + a.ice[1][c];
                if (a.ice[2]) t = t + a.ice[2][c];
                iwi = iwi + (t * dp) / gravity;
            }
        }
        pk = pk1;
    }
    const int o = i + d.nx * j;
    if (a.ivt) a.ivt[o] = ivt;
    if (a.iwv) a.iwv[o] = iwv;
    if (a.iwl) a.iwl[o] = iwl;
    if (a.iwi) a.iwi[o] = iwi;
}

struct ForceArgs { float *x[16]; const float *dq[16]; int stag[16]; int fb[16]; };

__global__ void __launch_bounds__(256)
k_apply_forcing(Dims d, ForceArgs a, double dt, int west, int east, int south, int north)
{
    const int m = blockIdx.z;
    const int nxm = d.nx + (a.stag[m] == 1), nym = d.ny + (a.stag[m] == 2);
    float *__restrict__ x = a.x[m];
    const float *__restrict__ dq = a.dq[m];
    if (!a.fb[m]) {                                             // whole field: u, v, w, pressure ...
        const size_t n = (size_t)nxm * d.nz * nym;
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
            x[t] = (float)((double)x[t] + ((double)dq[t] * dt));   // REAL + REAL*REAL(8)
        return;
    }
    // domain_obj.f90:2411-2423: only the true domain edges -- S/N full rows, W/E columns without the corner rows.
    // Enumerate just those cells: [south row | north row | west column | east column]
    const int nrow = nxm * d.nz, ncol = d.nz * (nym - 2);
    const int total = 2 * nrow + 2 * ncol;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int i, k, j; bool on;
        if (t < 2 * nrow) {
            const int r = t < nrow ? t : t - nrow;
            i = r % nxm; k = r / nxm; j = t < nrow ? 0 : nym - 1; on = t < nrow ? south : north;
        } else {
            const int s2 = t - 2 * nrow, r = s2 < ncol ? s2 : s2 - ncol;
            k = r % d.nz; j = 1 + r / d.nz; i = s2 < ncol ? 0 : nxm - 1; on = s2 < ncol ? west : east;
        }
        if (!on) continue;
        const size_t c = (size_t)i + (size_t)nxm * (k + (size_t)d.nz * j);
        x[c] = (float)((double)x[c] + ((double)dq[c] * dt));
    }
}

struct LimitArgs { float *x[16]; };
__global__ void __launch_bounds__(256)
k_enforce_limits(size_t n, LimitArgs a)
{
    float *x = a.x[blockIdx.y];
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
        if (x[t] < 0) x[t] = 0;
}
}  // namespace

// parts: 1 = thermodynamics, mass-point winds, column integrals (:60-144) ; 2 = w_real (:165-194) ; 3 = all of diagnostic_update.
// The sub-step splits 1 further: ICAR_DIAG_CELL (exner, T, density + the column integrals when they are on the device) before
// the microphysics, ICAR_DIAG_FACE (interface values, mass-point winds) beside it; 1 = both.  With column integrals on the
// device (they read the face kernel's outputs AND the water species) everything runs at ICAR_DIAG_CELL and _FACE is empty.
int icar_diagnostic_update_run(icar_hip_ctx *c, int parts)
{
    const float *u = (const float *)c->field[ICAR_F_U], *v = (const float *)c->field[ICAR_F_V];
    if (parts & 1) parts |= ICAR_DIAG_CELL | ICAR_DIAG_FACE;
    if (parts & (ICAR_DIAG_CELL | ICAR_DIAG_FACE)) {
    const float *p = icar_field_f(c, ICAR_F_PRESSURE), *th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE);
    if (!p || !th) return 1;
    float *ex = icar_field_f(c, ICAR_F_EXNER, false), *pi = icar_field_f(c, ICAR_F_PRESSURE_INTERFACE, false);
    float *ps = icar_field_f(c, ICAR_F_SURFACE_PRESSURE, false), *T = icar_field_f(c, ICAR_F_TEMPERATURE, false);
    float *Ti = icar_field_f(c, ICAR_F_TEMPERATURE_INTERFACE, false), *rho = icar_field_f(c, ICAR_F_DENSITY, false);
    if (!ex || !pi || !ps || !T || !Ti || !rho) return 1;
    float *um = u ? icar_field_f(c, ICAR_F_U_MASS, false) : nullptr, *vm = v ? icar_field_f(c, ICAR_F_V_MASS, false) : nullptr;
    ColumnArgs ca;
    ca.ivt = (float *)c->field[ICAR_F_IVT]; ca.iwv = (float *)c->field[ICAR_F_IWV];
    ca.iwl = (float *)c->field[ICAR_F_IWL]; ca.iwi = (float *)c->field[ICAR_F_IWI];
    const bool columns = ca.ivt || ca.iwv || ca.iwl || ca.iwi;
    ScopedTimer t(c, "diag");
    if (parts & ICAR_DIAG_CELL) {
        c->winds_valid = false;                                  // density is rewritten: Courant winds (advect_density) are stale -- u, v, w are
                                                                 // not: a prefetched CFL maximum of them stays valid (wind_version untouched)
        const size_t n4 = c->n3 / 4, rest = c->n3 - 4 * n4, nthr = n4 + rest;
        hipLaunchKernelGGL(k_diag_cell, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, c->stream, n4, c->n3, p, th, ex, T, rho);
    }
    if (columns ? (parts & ICAR_DIAG_CELL) : (parts & ICAR_DIAG_FACE)) {          // (with column integrals: at the CELL call)
        dim3 g((c->d.nx + 63) / 64, (c->d.nz + 3) / 4, c->d.ny), b(64, 4);
        hipLaunchKernelGGL(k_diag_face, g, b, 0, c->stream, c->d, p, T, pi, ps, Ti, u, v, um, vm);
    }
    if (columns && (parts & ICAR_DIAG_CELL)) {
        ca.qv = (const float *)c->field[ICAR_F_WATER_VAPOR]; ca.um = um; ca.vm = vm; ca.p_i = pi;
        ca.liq[0] = (const float *)c->field[ICAR_F_CLOUD_WATER]; ca.liq[1] = (const float *)c->field[ICAR_F_RAIN];
        ca.ice[0] = (const float *)c->field[ICAR_F_CLOUD_ICE]; ca.ice[1] = (const float *)c->field[ICAR_F_SNOW];
        ca.ice[2] = (const float *)c->field[ICAR_F_GRAUPEL];
        if ((ca.ivt || ca.iwv) && !ca.qv) { icar_set_error("diagnostic_update: ivt / iwv need water_vapor on the device"); return 1; }
        if (ca.ivt && (!um || !vm)) { icar_set_error("diagnostic_update: ivt needs u and v on the device"); return 1; }
        hipLaunchKernelGGL(k_diag_columns, dim3((c->d.nx + 63) / 64, c->d.ny), dim3(64), 0, c->stream, c->d, ca);
    }
    }
    if (!(parts & 2)) { HIPCHK(hipGetLastError()); return 0; }
    const float *w = (const float *)c->field[ICAR_F_W], *dzdx = (const float *)c->field[ICAR_F_DZDX];
    const float *dzdy = (const float *)c->field[ICAR_F_DZDY], *jaco = (const float *)c->field[ICAR_F_JACOBIAN];
    if (u && v && w && dzdx && dzdy && jaco) {
        float *wr = icar_field_f(c, ICAR_F_W_REAL, false);
        if (!wr) return 1;
        ScopedTimer t(c, "diag");
        if ((size_t)(c->d.nx + 1) * c->d.nz * (c->d.ny + 1) * sizeof(float) >= ((size_t)1 << 31)) { icar_set_error("diagnostic_update: a field of 2 GiB or more is not supported (32-bit buffer offsets)"); return 1; }
        dim3 g2((c->d.nx - 2 + 63) / 64, c->d.ny - 2), b2(64);
        hipLaunchKernelGGL(k_diag_wreal, g2, b2, 0, c->stream, c->d, u, v, w, dzdx, dzdy, jaco, wr);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int icar_apply_forcing_run(icar_hip_ctx *c, double dt, const int *fields, const int *fb, int n, int w, int e, int s, int nn)
{
    if (n <= 0) return 0;
    if (n > 16) { icar_set_error("apply_forcing: at most 16 fields per call"); return 1; }
    ForceArgs a;
    for (int m = 0; m < n; ++m) {
        const int f = fields[m];
        if (f < 0 || f >= ICAR_N_FIELDS || f == ICAR_F_PRECIPITATION || f == ICAR_F_SNOWFALL || f == ICAR_F_GRAUPEL_ACC || f == ICAR_F_SURFACE_PRESSURE) {
            icar_set_error("apply_forcing: only 3-D REAL(4) fields"); return 1;
        }
        a.x[m] = icar_field_f(c, f);
        if (!a.x[m]) return 1;
        if (!c->dqdt[f]) { icar_set_error("apply_forcing: dqdt of a listed field was never uploaded"); return 1; }
        a.dq[m] = c->dqdt[f];
        a.stag[m] = (f == ICAR_F_U || f == ICAR_F_JACOBIAN_U || f == ICAR_F_DZDX) ? 1 : (f == ICAR_F_V || f == ICAR_F_JACOBIAN_V || f == ICAR_F_DZDY) ? 2 : 0;
        a.fb[m] = fb[m];
        // the Courant winds of icar_hip_setup_winds are stale once u, v, w, density or a jacobian moved
        if (f == ICAR_F_U || f == ICAR_F_V || f == ICAR_F_W || (f >= ICAR_F_DENSITY && f <= ICAR_F_ADVECTION_DZ)) icar_winds_changed(c);
    }
    ScopedTimer t(c, "forcing");
    dim3 g(2048, 1, n), b(256);
    hipLaunchKernelGGL(k_apply_forcing, g, b, 0, c->stream, c->d, a, dt, w, e, s, nn);
    HIPCHK(hipGetLastError());
    return 0;
}

int icar_enforce_limits_run(icar_hip_ctx *c, const int *fields, int n)
{
    if (n <= 0) return 0;
    if (n > 16) { icar_set_error("enforce_limits: at most 16 fields per call"); return 1; }
    LimitArgs a;
    for (int m = 0; m < n; ++m) {
        if (fields[m] < 0 || fields[m] >= ICAR_N_ADVECTABLE) { icar_set_error("enforce_limits: advectable scalars only"); return 1; }
        a.x[m] = icar_field_f(c, fields[m]);
        if (!a.x[m]) return 1;
    }
    dim3 g(2048, n), b(256);
    hipLaunchKernelGGL(k_enforce_limits, g, b, 0, c->stream, c->n3, a);
    HIPCHK(hipGetLastError());
    return 0;
}
