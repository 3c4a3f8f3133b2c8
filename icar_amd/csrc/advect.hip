// icar_amd/csrc/advect.hip -- wind setup and the donor-cell scheme on gfx950 (rows A1, A2, A5 of SURVEY.md section 8).
//
// Reference algorithm: src/physics/advect.f90 (upwind) and the driver of src/physics/adv_mpdata.f90 (advect3d,
// :356-418).  MPDATA's corrective iterations are ONE fused kernel each (mpdata.hip); this file holds
//   * k_setup_winds: U_m, V_m, W_m and W_m/dz (advect.f90:345-348, adv_mpdata.f90:500-506, :379);
//   * k_upwind_pass: the donor-cell pass of the upwind scheme (and of mpdata_order = 1), lanes along i (the contiguous
//     axis), winds / jacobian / density loaded once per thread and reused for all advected scalars of the
//     batch (algorithmic traffic 8N+16 B/cell).  HBM-bound (4.4 TB/s of traffic), so it keeps the reference's operation
//     order and IEEE division: bit-identical to the CPU reference (-ffp-contract=off);
//   * icar_advect_run: the sequencing of advect3d; advected scalars ping-pong between two device buffers, so no
//     copy-back pass exists.
#include "ctx.h"
#include <vector>
#include <algorithm>
#include <cstring>

#define BX 64
#define BY 4

__device__ __forceinline__ float flux1(float l, float r, float U)
{   // donor-cell flux, src/physics/adv_mpdata.f90:40
    return ((U + fabsf(U)) * l + (U - fabsf(U)) * r) / 2;
}

// ------------------------------------------------------------------------------------------------
// A1: U_m, V_m, W_m (+ W_m/dz used by the pseudo-velocity step, adv_mpdata.f90:379)
// ------------------------------------------------------------------------------------------------
// XCD-aware block -> tile mapping.  Workgroups are handed to the 8 XCDs round-robin in linear-id order, and each XCD
// has its own 4 MB L2.  Stencil kernels re-read the neighbouring j / k planes of the block next to them; with the
// default mapping that neighbour runs on another XCD and every overlap plane is fetched from HBM again by each of
// them.  Here XCD x works on the contiguous range [x*n/8, (x+1)*n/8) of the (i fastest, then k, then j) tile order, so
// the blocks that share planes run on the same XCD close together in time and the overlap hits in L2.
struct TileId { int x, y, z; };
// rows: how many z-slices (j rows / j slabs) of tiles one XCD takes before the next XCD's turn.  A single contiguous
// range per XCD (rows = gz/8) would share the most planes but hands each XCD one region of the domain -- clouds are not
// spread evenly, and the launch then lasts as long as the cloudiest eighth.
__device__ __forceinline__ TileId xcd_tile(unsigned rows)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned chunk = gx * gy * rows, super = chunk * 8;
    const unsigned s = id / super, w = id % super;
    if ((s + 1) * super <= n) id = s * super + (w % 8) * chunk + w / 8;      // the last partial super-chunk keeps its order
    TileId t;
    t.x = (int)(id % gx); t.y = (int)((id / gx) % gy); t.z = (int)(id / (gx * gy));
    return t;
}

// WREAL: also w_real of diagnostic_update (time_step.f90:165-194) for the interior columns -- the same expression and operand order
// as k_diag_wreal (step.hip), from the winds this kernel reads anyway.  The sub-step asks for it (timestep.hip): beside the
// advection those 0.3 GB of reads cost the latency-bound MPDATA kernel 40 us, here they add three static arrays to a streaming kernel
// beside the microphysics.
template <int SCHEME, bool RHO, bool WREAL>
__global__ void __launch_bounds__(BX * BY)
k_setup_winds(Dims d, const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w,
              const float *__restrict__ rho, const float *__restrict__ ju, const float *__restrict__ jv,
              const float *__restrict__ jw, const float *__restrict__ dz, float dt, float dx,
              float *__restrict__ U, float *__restrict__ V, float *__restrict__ W, float *__restrict__ Wdz,
              const float *__restrict__ dzdx, const float *__restrict__ dzdy, const float *__restrict__ jaco, float *__restrict__ w_real)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + threadIdx.y;
    const int j = tb.z;
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    const float r0 = RHO ? rho[c] : 1.0f;
    float Uv = 0.0f, Vv = 0.0f, Wv;
    const int cu = i + (d.nx + 1) * (k + d.nz * j);
    const float uc = u[cu], vc = v[c], wc = w[c];
    if (i >= 1) {
        const float rl = RHO ? rho[c - 1] : 1.0f;
        if (SCHEME == 1) Uv = uc * dt * ju[cu] * (r0 + rl) * 0.5f / dx;      // advect.f90:345
        else             Uv = uc * dt * (r0 + rl) * 0.5f * ju[cu] / dx;      // adv_mpdata.f90:500
    }
    if (j >= 1) {
        const float rl = RHO ? rho[c - d.sj] : 1.0f;
        if (SCHEME == 1) Vv = vc * dt * jv[c] * (r0 + rl) * 0.5f / dx;
        else             Vv = vc * dt * (r0 + rl) * 0.5f * jv[c] / dx;
    }
    if (k < d.nz - 1) {
        const float ru = RHO ? rho[c + d.sk] : 1.0f;
        Wv = wc * dt * jw[c] * (ru + r0) * 0.5f;
    } else
        Wv = wc * dt * jw[c] * r0;
    U[c] = Uv; V[c] = Vv; W[c] = Wv; Wdz[c] = Wv / dz[c];
    if (WREAL && i >= 1 && i < d.nx - 1 && j >= 1 && j < d.ny - 1) {
        const float uw0 = uc * dzdx[cu], uw1 = u[cu + 1] * dzdx[cu + 1];                    // uw(i), uw(i+1)
        const float vw0 = vc * dzdy[c], vw1 = v[c + d.sj] * dzdy[c + d.sj];                // vw(j), vw(j+1)
        const float lastw = (k > 0) ? w[c - d.sk] : 0.0f;
        w_real[c] = (uw0 + uw1) * 0.5f + (vw0 + vw1) * 0.5f + jaco[c] * (lastw + wc) * 0.5f;
    }
}

// ------------------------------------------------------------------------------------------------
// A2: donor-cell pass with winds shared by all scalars (advect.f90:139-175, adv_mpdata.f90:44-105)
// ------------------------------------------------------------------------------------------------
template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_upwind_pass(Dims d, CVarPtrs in, VarPtrs out, int nv,
              const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ W,
              const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + threadIdx.y;
    const int j = tb.z;
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    const bool interior = (i > 0) && (i < d.nx - 1) && (j > 0) && (j < d.ny - 1);
    if (!interior) {
        for (int m = 0; m < nv; ++m) out.p[m][c] = in.p[m][c];
        return;
    }
    const float Ur = U[c + 1], Ul = U[c], Vn = V[c + d.sj], Vs = V[c], Wt = W[c];
    const float Wb = (k > 0) ? W[c - d.sk] : 0.0f;
    const float r = RHO ? rho[c] : 1.0f;
    const float ja = jaco[c];
    const float den_h = ja * r;
    const float den_v = dz[c] * ja * r;
    const bool bottom = (k == 0), top = (k == d.nz - 1);
    for (int m = 0; m < nv; ++m) {
        const float *__restrict__ q = in.p[m];
        const float q0 = q[c];
        const float f1r = flux1(q0, q[c + 1], Ur);
        const float f1l = flux1(q[c - 1], q0, Ul);
        const float f3 = flux1(q0, q[c + d.sj], Vn);
        const float f4 = flux1(q[c - d.sj], q0, Vs);
        float qq = q0 - ((f1r - f1l) + (f3 - f4)) / den_h;
        if (bottom) {
            qq = qq - flux1(q0, q[c + d.sk], Wt) / den_v;
        } else if (top) {
            qq = qq - (q0 * Wt - flux1(q[c - d.sk], q0, Wb)) / den_v;
        } else {
            qq = qq - (flux1(q0, q[c + d.sk], Wt) - flux1(q[c - d.sk], q0, Wb)) / den_v;
        }
        out.p[m][c] = qq;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static dim3 grid3(const Dims &d) { return dim3((d.nx + BX - 1) / BX, (d.nz + BY - 1) / BY, d.ny); }

int icar_mpdata_fused_run(icar_hip_ctx *c, bool rho_on, bool fct, bool pass1, const CVarPtrs &in, const VarPtrs &out, int nv);   // mpdata.hip
int icar_mpdata_coef_run(icar_hip_ctx *c, bool rho_on);                                                                       // mpdata.hip
int icar_mpdata_exact_run(icar_hip_ctx *c, bool rho_on, bool fct, int order, const CVarPtrs &q, const VarPtrs &alt, int nv);   // mpdata_exact.hip

// donor-cell pass in -> out (distinct buffers) with the given face velocities (U_m, V_m, W_m, or one scalar's limited pseudo-velocities)
int icar_upwind_pass_run(icar_hip_ctx *c, bool rho_on, const CVarPtrs &in, const VarPtrs &out, int nv,
                         const float *U, const float *V, const float *W)
{
    const float *rho = rho_on ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    if (!jaco || !dz || (rho_on && !rho)) return 1;
    dim3 g = grid3(c->d), b(BX, BY);
    if (rho_on) hipLaunchKernelGGL((k_upwind_pass<true>), g, b, 0, c->stream, c->d, in, out, nv, U, V, W, rho, jaco, dz);
    else        hipLaunchKernelGGL((k_upwind_pass<false>), g, b, 0, c->stream, c->d, in, out, nv, U, V, W, rho, jaco, dz);
    HIPCHK(hipGetLastError());
    return 0;
}

static int ensure_adv_scratch(icar_hip_ctx *c)
{
    const size_t bytes = c->n3 * sizeof(float);
    if (!c->U) {
        HIPCHK(hipMalloc(&c->U, bytes)); HIPCHK(hipMalloc(&c->V, bytes));
        HIPCHK(hipMalloc(&c->W, bytes)); HIPCHK(hipMalloc(&c->Wdz, bytes));
    }
    return 0;
}

// with_wreal: also w_real (diagnostic_update part 2) from the same winds; *wreal_done says whether the fields for it were there
int icar_advect_setup_winds(icar_hip_ctx *c, int scheme, float dt, float dx, int advect_density, bool with_wreal, bool *wreal_done)
{
    if (wreal_done) *wreal_done = false;
    if (scheme != ICAR_ADV_UPWIND && scheme != ICAR_ADV_MPDATA) { icar_set_error("setup_winds: bad scheme"); return 1; }
    if (ensure_adv_scratch(c)) return 1;
    const float *u = icar_field_f(c, ICAR_F_U), *v = icar_field_f(c, ICAR_F_V), *w = icar_field_f(c, ICAR_F_W);
    const float *ju = icar_field_f(c, ICAR_F_JACOBIAN_U), *jv = icar_field_f(c, ICAR_F_JACOBIAN_V);
    const float *jw = icar_field_f(c, ICAR_F_JACOBIAN_W), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    const float *rho = advect_density ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    if (!u || !v || !w || !ju || !jv || !jw || !dz || (advect_density && !rho)) return 1;
    const float *dzdx = nullptr, *dzdy = nullptr, *jaco = nullptr; float *wr = nullptr;
    if (with_wreal) {
        dzdx = (const float *)c->field[ICAR_F_DZDX]; dzdy = (const float *)c->field[ICAR_F_DZDY]; jaco = (const float *)c->field[ICAR_F_JACOBIAN];
        if (dzdx && dzdy && jaco) { wr = icar_field_f(c, ICAR_F_W_REAL, false); if (!wr) return 1; }
        else with_wreal = false;                         // (diagnostic_update leaves w_real alone without them, too)
    }
    ScopedTimer t(c, "winds");
    dim3 g = grid3(c->d), b(BX, BY);
#define LAUNCH(S, R, WR) hipLaunchKernelGGL((k_setup_winds<S, R, WR>), g, b, 0, c->stream, c->d, u, v, w, rho, ju, jv, jw, dz, dt, dx, c->U, c->V, c->W, c->Wdz, dzdx, dzdy, jaco, wr)
#define LAUNCH2(S, R) { if (with_wreal) LAUNCH(S, R, true); else LAUNCH(S, R, false); }
    if (scheme == 1) { if (advect_density) LAUNCH2(1, true) else LAUNCH2(1, false) }
    else             { if (advect_density) LAUNCH2(2, true) else LAUNCH2(2, false) }
#undef LAUNCH2
#undef LAUNCH
    if (wreal_done) *wreal_done = with_wreal;
    HIPCHK(hipGetLastError());
    // MPDATA: everything of the corrective iteration that does not depend on the scalar, once per step (mpdata.hip)
    if (scheme == ICAR_ADV_MPDATA && icar_mpdata_coef_run(c, advect_density != 0)) return 1;
    c->winds_valid = true;
    c->step.winds_scheme = scheme; c->step.winds_dt = dt; c->step.winds_dens = advect_density ? 1 : 0;   // what advect() of the step driver asks for
    return 0;
}

int icar_advect_run(icar_hip_ctx *c, int scheme, int order, int fct, int advect_density, const int *fields, int n)
{
    if (!c->winds_valid) { icar_set_error("advect: call icar_hip_setup_winds first (after every change of u, v, w, density or the jacobians)"); return 1; }
    if (n <= 0) return 0;
    if (n > ICAR_MAX_ADV) { icar_set_error("advect: too many fields"); return 1; }
    if (scheme == ICAR_ADV_UPWIND) order = 1;
    if (order < 1) { icar_set_error("advect: mpdata_order must be >= 1"); return 1; }
    if (ensure_adv_scratch(c)) return 1;
    const float *rho = advect_density ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    if (!jaco || !dz || (advect_density && !rho)) return 1;
    CVarPtrs q; VarPtrs alt;
    for (int m = 0; m < n; ++m) {
        const int f = fields[m];
        if (f < 0 || f >= ICAR_N_ADVECTABLE) { icar_set_error("advect: field id is not an advectable scalar"); return 1; }
        for (int mm = 0; mm < m; ++mm) if (fields[mm] == f) { icar_set_error("advect: duplicate field"); return 1; }
        float *p = icar_field_f(c, f);
        if (!p) return 1;
        if (!c->alt[f]) HIPCHK(hipMalloc(&c->alt[f], c->n3 * sizeof(float)));
        q.p[m] = p; alt.p[m] = c->alt[f];
    }
    ScopedTimer t(c, "advect");
    auto swap_fields = [&]() {
        for (int m = 0; m < n; ++m) {
            const int f = fields[m];
            float *cur = (float *)c->field[f];
            c->field[f] = c->alt[f]; c->alt[f] = cur;
            q.p[m] = (float *)c->field[f]; alt.p[m] = c->alt[f];
        }
    };
    if (order == 1) {
        // q -> alt, swap  (upwind; or mpdata_order=1: adv_mpdata.f90:374,404-411)
        if (icar_upwind_pass_run(c, advect_density != 0, q, alt, n, c->U, c->V, c->W)) return 1;
        swap_fields();
        return 0;
    }
    if (c->mpdata_exact) {
        // the reference's own operation order, four launches per scalar (mpdata_exact.hip): bit-identical to the CPU reference
        if (icar_mpdata_exact_run(c, advect_density != 0, fct != 0, order, q, alt, n)) return 1;
        swap_fields();
        return 0;
    }
    // adv_mpdata.f90:372-402: iord = 1 (donor cell) + iord = 2 (pseudo-velocities from the pass-1 field, limiter against the
    // field before the step, donor cell with the limited pseudo-velocities) are one fused launch; every further corrective
    // iteration starts from q2 = q (:393-402) with the ORIGINAL U_m, V_m, W_m/dz (:379) and is the same launch without
    // the first donor-cell pass.
    for (int iord = 2; iord <= order; ++iord) {
        if (icar_mpdata_fused_run(c, advect_density != 0, fct != 0, iord == 2, q, alt, n)) return 1;
        swap_fields();
    }
    return 0;
}
