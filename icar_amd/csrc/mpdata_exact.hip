// icar_amd/csrc/mpdata_exact.hip -- MPDATA in the REFERENCE'S OWN OPERATION ORDER (icar_hip_mpdata_exact(ctx, 1)): every sum,
// product and quotient of adv_mpdata.f90 evaluated as written (IEEE division, no contraction, the C library's tie rule for
// max / min), so that the advected fields are bit-identical to the CPU reference and a whole sub-step sequence (microphysics ->
// halo -> advection -> forcing) can be compared bit for bit over many steps.  The fused kernel of mpdata.hip stays the default:
// it reads and writes each scalar once; this path keeps the reference's four stages as four launches per scalar and exchanges
// q2, u2, v2, w2 through HBM (~100 B per scalar-cell instead of 8).
//
//   donor cell   q  -> q2                      adv_mpdata.f90:44-105, :374      k_upwind_pass (advect.hip, bit-exact since round 1)
//   velocities   q2 -> u2, v2, w2 (* 0.5, dz)  adv_mpdata.f90:107-255, :383-385 k_mpx_velocities
//   limiter      q, q2, u2.. -> limited u2..   adv_mpdata.f90:257-354 + adv_mpdata_FCT_core.f90:47-116   k_mpx_limit
//   donor cell   q2 -> q with the limited pseudo-velocities   :389              k_upwind_pass
//   mpdata_order > 2: q2 := q and again from "velocities" with the original U_m, V_m, W_m/dz (:379, :393-402)
//
// The limiter's carried variables (qmax_i, qmax_i2, fin_i, ...: adv_mpdata_FCT_core.f90 walks a line and hands "i2" of one face
// to "i" of the next) are functions of a face's three nearest unlimited fluxes and the four cells around it, so faces are
// independent: one thread per cell limits its x, y and z face, and the limited velocities go to their own arrays because the
// neighbours' fluxes are those of the UNLIMITED velocities (the f(:) of flux1 is computed before the line is walked, :47).
#include "ctx.h"
#include <algorithm>

#define BX 64
#define BY 4

int icar_upwind_pass_run(icar_hip_ctx *c, bool rho_on, const CVarPtrs &in, const VarPtrs &out, int nv,
                         const float *U, const float *V, const float *W);                                   // advect.hip

namespace {

// glibc's fmaxf / fminf on x86-64 are maxss / minss after a NaN test: of two equal arguments (+0 and -0 included) the SECOND
// is returned.  v_max_f32 orders -0 < +0; written as the comparison so that the bits agree.
__device__ __forceinline__ float o_max(float x, float y) { return x > y ? x : y; }
__device__ __forceinline__ float o_min(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float max4(float a, float b, float c, float d) { return o_max(o_max(o_max(a, b), c), d); }
__device__ __forceinline__ float min4(float a, float b, float c, float d) { return o_min(o_min(o_min(a, b), c), d); }

__device__ __forceinline__ float flux1(float l, float r, float U)
{   // adv_mpdata.f90:40
    return ((U + fabsf(U)) * l + (U - fabsf(U)) * r) / 2;
}

// ------------------------------------------------------------------------------------------------
// anti-diffusive pseudo-velocities of one scalar (mpdata_fluxes, adv_mpdata.f90:107-255) and their scaling (:383-385)
// u2(c): face between i-1 and i; v2(c): between j-1 and j; w2(c): above level k.  u, v = U_m, V_m; w = W_m / dz.
// ------------------------------------------------------------------------------------------------
template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_mpx_velocities(Dims d, const float *__restrict__ q, const float *__restrict__ u, const float *__restrict__ v,
                 const float *__restrict__ w, const float *__restrict__ rho, const float *__restrict__ jaco,
                 const float *__restrict__ dz, float *__restrict__ u2, float *__restrict__ v2, float *__restrict__ w2)
{
    const int i = blockIdx.x * BX + threadIdx.x;
    const int k = blockIdx.y * BY + threadIdx.y;
    const int j = blockIdx.z;
    if (i >= d.nx || k >= d.nz) return;
    const int nx = d.nx, nz = d.nz, ny = d.ny;
    const int c = d.idx(i, k, j);
#define Q(ii, kk, jj) q[d.idx(ii, kk, jj)]
#define G(cc) (jaco[cc] * (RHO ? rho[cc] : 1.0f))
    float ru = 0.0f, rv = 0.0f, rw = 0.0f;
    if (i >= 1) {                                                                               // :134-169
        const float rx = Q(i, k, j), lx = Q(i - 1, k, j);
        const float denomx = (rx + lx + 1e-10f);
        const float Gs = G(c) + G(c - 1);
        float val = fabsf(u[c]) * (1 - fabsf(u[c]) / (0.5f * Gs));
        val = val * (rx - lx) / denomx;
        if (j > 0 && j < ny - 1) {                                                              // UxV
            const float eq = (Q(i, k, j + 1) - Q(i, k, j - 1) + Q(i - 1, k, j + 1) - Q(i - 1, k, j - 1)) /
                             (Q(i, k, j + 1) + Q(i, k, j - 1) + Q(i - 1, k, j + 1) + Q(i - 1, k, j - 1) + 1e-10f);
            const float ev = (1 / 4.0f) * (v[c] + v[c + d.sj] + v[c - 1] + v[c - 1 + d.sj]);
            val = val - 0.5f * u[c] * ev * eq / Gs;
        }
        if (k > 0 && k < nz - 1) {                                                              // UxW
            const float eq = (Q(i, k + 1, j) - Q(i, k - 1, j) + Q(i - 1, k + 1, j) - Q(i - 1, k - 1, j)) /
                             (Q(i, k + 1, j) + Q(i, k - 1, j) + Q(i - 1, k + 1, j) + Q(i - 1, k - 1, j) + 1e-10f);
            const float ev = (1 / 4.0f) * (w[c] + w[c - d.sk] + w[c - 1] + w[c - 1 - d.sk]);
            val = val - 0.5f * u[c] * ev * eq / Gs;
        }
        ru = val;
    }
    if (j >= 1) {                                                                               // :172-208
        const float r = Q(i, k, j), l = Q(i, k, j - 1);
        const float denom = (r + l + 1e-10f);
        const float Gs = G(c) + G(c - d.sj);
        float val = fabsf(v[c]) * (1 - fabsf(v[c]) / (0.5f * Gs));
        val = val * (r - l) / denom;
        {                                                                                       // VxU (zero in the x ring)
            float eq = 0, ev = 0;
            if (i > 0 && i < nx - 1) {
                eq = (Q(i + 1, k, j - 1) - Q(i - 1, k, j) + Q(i + 1, k, j) - Q(i - 1, k, j - 1)) /
                     (Q(i + 1, k, j) + Q(i + 1, k, j - 1) + Q(i - 1, k, j) + Q(i - 1, k, j - 1) + 1e-10f);
                ev = (1 / 4.0f) * (u[c + 1] + u[c + 1 - d.sj] + u[c] + u[c - d.sj]);
            }
            val = val - 0.5f * v[c] * ev * eq / Gs;
        }
        if (k > 0 && k < nz - 1) {                                                              // VxW
            const float eq = (Q(i, k + 1, j - 1) - Q(i, k - 1, j) + Q(i, k + 1, j) - Q(i, k - 1, j - 1)) /
                             (Q(i, k + 1, j - 1) + Q(i, k - 1, j) + Q(i, k + 1, j) + Q(i, k - 1, j - 1) + 1e-10f);
            const float ev = (1 / 4.0f) * (w[c] + w[c - d.sk] + w[c - d.sj] + w[c - d.sk - d.sj]);
            val = val - 0.5f * v[c] * ev * eq / Gs;
        }
        rv = val;
    }
    if (k < nz - 1) {                                                                           // :214-249
        const float r = Q(i, k + 1, j), l = Q(i, k, j);
        const float denom = (r + l + 1e-10f);
        const float Gs = G(c + d.sk) + G(c);
        float val = fabsf(w[c]) * (1 - fabsf(w[c]) / (0.5f * Gs));
        val = val * (r - l) / denom;
        {                                                                                       // WxU
            float eq = 0, ev = 0;
            if (i > 0 && i < nx - 1) {
                eq = (Q(i + 1, k + 1, j) - Q(i - 1, k, j) + Q(i + 1, k, j) - Q(i - 1, k + 1, j)) /
                     (Q(i + 1, k, j) + Q(i + 1, k + 1, j) + Q(i - 1, k, j) + Q(i - 1, k + 1, j) + 1e-10f);
                ev = (1 / 4.0f) * (u[c + 1] + u[c + 1 + d.sk] + u[c] + u[c + d.sk]);
            }
            val = val - 0.5f * w[c] * ev * eq / Gs;
        }
        if (j > 0 && j < ny - 1) {                                                              // WxV
            const float eq = (Q(i, k + 1, j + 1) - Q(i, k, j - 1) + Q(i, k, j + 1) - Q(i, k + 1, j - 1)) /
                             (Q(i, k, j + 1) + Q(i, k + 1, j - 1) + Q(i, k + 1, j + 1) + Q(i, k, j - 1) + 1e-10f);
            const float ev = (1 / 4.0f) * (v[c] + v[c + d.sk] + v[c + d.sj] + v[c + d.sk + d.sj]);
            val = val - 0.5f * w[c] * ev * eq / Gs;
        }
        rw = val;
    }
#undef Q
#undef G
    u2[c] = ru * 0.5f;                                                                           // :383-385
    v2[c] = rv * 0.5f;
    w2[c] = rw * 0.5f * dz[c];
}

// ------------------------------------------------------------------------------------------------
// one face of the flux-corrected-transport limiter (adv_mpdata_FCT_core.f90:47-116): face t of a line of n cells lies between
// cells t ("c") and t+1 ("p"); "m" = cell t-1, "pp" = cell t+2; Um, Uc, Up = unlimited velocities of faces t-1, t, t+1.
// Returns the limited velocity of face t.  Arguments that do not exist for this t (m at t = 0, pp / Up at t = n-2) are not used.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fct_face(int t, int n, bool is_w, float q1m, float q1c, float q1p, float q1pp,
                                          float lm, float lc, float lp, float lpp, float Um, float Uc, float Up)
{
    const float fc = flux1(q1c, q1p, Uc);
    float qmax_i, qmin_i, qmax_i2, qmin_i2, fin_i, fout_i, fin_i2, fout_i2;
    if (t == 0) {
        qmax_i = max4(q1c, q1p, lc, lp);
        qmin_i = min4(q1c, q1p, lc, lp);
        if (is_w) { fin_i = 0.f - o_min(0.f, fc); fout_i = o_max(0.f, fc); }
        else      { fin_i = 0; fout_i = 0; }
    } else {                                            // "i2" of face t-1
        const float fm = flux1(q1m, q1c, Um);
        qmax_i = o_max(max4(q1m, q1c, q1p, lm), o_max(lc, lp));
        qmin_i = o_min(min4(q1m, q1c, q1p, lm), o_min(lc, lp));
        fin_i  = o_max(0.f, fm) - o_min(0.f, fc);
        fout_i = o_max(0.f, fc) - o_min(0.f, fm);
    }
    if (t != n - 2) {
        const float fp = flux1(q1p, q1pp, Up);
        qmax_i2 = o_max(max4(q1c, q1p, q1pp, lc), o_max(lp, lpp));
        qmin_i2 = o_min(min4(q1c, q1p, q1pp, lc), o_min(lp, lpp));
        fin_i2  = o_max(0.f, fc) - o_min(0.f, fp);
        fout_i2 = o_max(0.f, fp) - o_min(0.f, fc);
    } else {
        qmax_i2 = o_max(o_max(q1c, q1p), lc);
        qmin_i2 = o_min(o_min(q1c, q1p), lc);
        if (is_w) { fin_i2 = o_max(0.f, fc) - o_min(0.f, fc); fout_i2 = o_max(0.f, fc) - o_min(0.f, fc); }
        else      { fin_i2 = 0; fout_i2 = 0; }
    }
    float U = Uc;
    if (Uc > 0) {
        const float beta_out_i = (q1c - qmin_i) / (fout_i + 1e-15f);
        const float beta_in_i2 = (qmax_i2 - q1p) / (fin_i2 + 1e-15f);
        U = o_min(o_min(1.f, beta_in_i2), beta_out_i) * Uc;
    } else if (Uc < 0) {
        const float beta_in_i = (qmax_i - q1c) / (fin_i + 1e-15f);
        const float beta_out_i2 = (q1p - qmin_i2) / (fout_i2 + 1e-15f);
        U = o_min(o_min(1.f, beta_in_i), beta_out_i2) * Uc;
    }
    return U;
}

// flux_limiter (adv_mpdata.f90:257-354): x lines and z lines of the rows j = 2 .. ny-1 (z lines of the columns i = 2 .. nx-1 only),
// y lines of every (i, k).  l = the field the iteration started from, q1 = after the donor-cell pass.
__global__ void __launch_bounds__(BX * BY)
k_mpx_limit(Dims d, const float *__restrict__ l, const float *__restrict__ q1,
            const float *__restrict__ u2, const float *__restrict__ v2, const float *__restrict__ w2,
            float *__restrict__ u2l, float *__restrict__ v2l, float *__restrict__ w2l)
{
    const int i = blockIdx.x * BX + threadIdx.x;
    const int k = blockIdx.y * BY + threadIdx.y;
    const int j = blockIdx.z;
    if (i >= d.nx || k >= d.nz) return;
    const int nx = d.nx, nz = d.nz, ny = d.ny;
    const int c = d.idx(i, k, j);
    const bool row = (j > 0) && (j < ny - 1);
    // x face between i-1 and i: t = i-1 of a line of nx cells
    float ru = u2[c];
    if (row && i >= 1) {
        const int t = i - 1;
        const bool hm = t > 0, hp = t != nx - 2;
        ru = fct_face(t, nx, false, hm ? q1[c - 2] : 0.f, q1[c - 1], q1[c], hp ? q1[c + 1] : 0.f,
                      hm ? l[c - 2] : 0.f, l[c - 1], l[c], hp ? l[c + 1] : 0.f,
                      hm ? u2[c - 1] : 0.f, u2[c], hp ? u2[c + 1] : 0.f);
    }
    u2l[c] = ru;
    // y face between j-1 and j: t = j-1 of a line of ny cells
    float rv = v2[c];
    if (j >= 1) {
        const int t = j - 1, s = d.sj;
        const bool hm = t > 0, hp = t != ny - 2;
        rv = fct_face(t, ny, false, hm ? q1[c - 2 * s] : 0.f, q1[c - s], q1[c], hp ? q1[c + s] : 0.f,
                      hm ? l[c - 2 * s] : 0.f, l[c - s], l[c], hp ? l[c + s] : 0.f,
                      hm ? v2[c - s] : 0.f, v2[c], hp ? v2[c + s] : 0.f);
    }
    v2l[c] = rv;
    // z face above level k: t = k of a line of nz cells; w(kme) = 0 afterwards (:322)
    float rw = w2[c];
    if (row && i > 0 && i < nx - 1) {
        if (k == nz - 1) rw = 0.f;
        else {
            const int t = k, s = d.sk;
            const bool hm = t > 0, hp = t != nz - 2;
            rw = fct_face(t, nz, true, hm ? q1[c - s] : 0.f, q1[c], q1[c + s], hp ? q1[c + 2 * s] : 0.f,
                          hm ? l[c - s] : 0.f, l[c], l[c + s], hp ? l[c + 2 * s] : 0.f,
                          hm ? w2[c - s] : 0.f, w2[c], hp ? w2[c + s] : 0.f);
        }
    }
    w2l[c] = rw;
}

}   // namespace

// advect3d (adv_mpdata.f90:356-418) for every scalar of the batch, mpdata_order >= 2: q[m] -> alt[m] (the caller swaps)
int icar_mpdata_exact_run(icar_hip_ctx *c, bool rho_on, bool fct, int order, const CVarPtrs &q, const VarPtrs &alt, int nv)
{
    const Dims &d = c->d;
    if (d.nx < 3 || d.ny < 3 || d.nz < 3) { icar_set_error("mpdata (exact): the limiter needs lines of at least 3 cells (nx, ny, nz >= 3)"); return 1; }
    const float *rho = rho_on ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    if (!jaco || !dz || (rho_on && !rho)) return 1;
    if (!c->mpx_buf) HIPCHK(hipMalloc(&c->mpx_buf, 8 * c->n3 * sizeof(float)));
    float *A = c->mpx_buf, *B = A + c->n3, *u2 = B + c->n3, *v2 = u2 + c->n3, *w2 = v2 + c->n3;
    float *u2l = w2 + c->n3, *v2l = u2l + c->n3, *w2l = v2l + c->n3;
    const dim3 g((d.nx + BX - 1) / BX, (d.nz + BY - 1) / BY, d.ny), b(BX, BY);
    for (int m = 0; m < nv; ++m) {
        CVarPtrs in1; VarPtrs out1;
        in1.p[0] = q.p[m]; out1.p[0] = A;
        if (icar_upwind_pass_run(c, rho_on, in1, out1, 1, c->U, c->V, c->W)) return 1;             // iord = 1 (:374)
        float *q2 = A, *spare = B;
        for (int iord = 2; iord <= order; ++iord) {
            if (rho_on) hipLaunchKernelGGL((k_mpx_velocities<true>), g, b, 0, c->stream, d, q2, c->U, c->V, c->Wdz, rho, jaco, dz, u2, v2, w2);
            else        hipLaunchKernelGGL((k_mpx_velocities<false>), g, b, 0, c->stream, d, q2, c->U, c->V, c->Wdz, rho, jaco, dz, u2, v2, w2);
            const float *l = (iord == 2) ? q.p[m] : q2;                                           // :393-402: from iord = 3 on q == q2
            if (fct) hipLaunchKernelGGL(k_mpx_limit, g, b, 0, c->stream, d, l, q2, u2, v2, w2, u2l, v2l, w2l);
            HIPCHK(hipGetLastError());
            float *dst = (iord == order) ? alt.p[m] : spare;
            in1.p[0] = q2; out1.p[0] = dst;
            if (icar_upwind_pass_run(c, rho_on, in1, out1, 1, fct ? u2l : u2, fct ? v2l : v2, fct ? w2l : w2)) return 1;   // :389
            spare = q2; q2 = dst;
        }
    }
    return 0;
}
