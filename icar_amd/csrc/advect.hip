// icar_amd/csrc/advect.hip -- scalar advection on gfx950 (rows A1-A5 of SURVEY.md section 8).
//
// Reference algorithm: src/physics/advect.f90 (upwind), src/physics/adv_mpdata.f90 +
// adv_mpdata_FCT_core.f90 (MPDATA order 2 with flux-corrected transport).  Written from the
// algorithm, not from the Fortran loop structure:
//   * lanes run along i (the contiguous axis) so every global access is coalesced;
//   * the Courant-number winds / jacobian / density are loaded once per thread and reused for all
//     advected scalars of the batch (algorithmic traffic 8N+16 B/cell for the donor-cell pass);
//   * the FCT limiter is never materialised: the final donor-cell pass limits its own six face
//     pseudo-velocities on the fly (every face of adv_mpdata_FCT_core.f90 is independent given
//     the unlimited fluxes), which removes a 12 B/cell write + 12 B/cell read per scalar;
//   * advected scalars ping-pong between two device buffers so no copy-back pass exists.
// Arithmetic follows the reference's operation order; compiled with -ffp-contract=off and
// IEEE division so the result is bit-identical to the CPU reference.
#include "ctx.h"
#include <vector>
#include <algorithm>
#include <cstring>

#define BX 64
#define BY 4

// Every quotient of the scheme goes through fdiv() = the compiler's IEEE expansion (v_div_scale x2, v_rcp, fma, fma, mul,
// fma, fma, fma, v_div_fmas, v_div_fixup).  Tried and dropped (profiles/micro/divbench.hip):
//  * the bare rcp / fma / mul chain on a 2^64-scaled numerator (Markstein): bit-identical for normal quotients and 3-6 %
//    faster per kernel, but it rounds twice when the quotient is subnormal -- and the tails of the hydrometeor fields do
//    reach 1e-39 (one cell of tests/test_gpu_advect.py::test_mpdata_sparse_fields... differed by 1.4e-45);
//  * a per-wave "all operands safe ? lean : IEEE" branch: slower than the IEEE division alone, because it cuts the basic
//    block at every quotient and the independent divisions no longer interleave.
#ifdef ICAR_EXPERIMENT_RCP
__device__ __forceinline__ float fdiv(float n, float d) { return n * __builtin_amdgcn_rcpf(d); }
#else
__device__ __forceinline__ float fdiv(float n, float d) { return n / d; }
#endif

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

template <int SCHEME, bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_setup_winds(Dims d, const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w,
              const float *__restrict__ rho, const float *__restrict__ ju, const float *__restrict__ jv,
              const float *__restrict__ jw, const float *__restrict__ dz, float dt, float dx,
              float *__restrict__ U, float *__restrict__ V, float *__restrict__ W, float *__restrict__ Wdz)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + threadIdx.y;
    const int j = tb.z;
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    const float r0 = RHO ? rho[c] : 1.0f;
    float Uv = 0.0f, Vv = 0.0f, Wv;
    if (i >= 1) {
        const float rl = RHO ? rho[c - 1] : 1.0f;
        const int cu = i + (d.nx + 1) * (k + d.nz * j);
        if (SCHEME == 1) Uv = u[cu] * dt * ju[cu] * (r0 + rl) * 0.5f / dx;      // advect.f90:345
        else             Uv = u[cu] * dt * (r0 + rl) * 0.5f * ju[cu] / dx;      // adv_mpdata.f90:500
    }
    if (j >= 1) {
        const float rl = RHO ? rho[c - d.sj] : 1.0f;
        if (SCHEME == 1) Vv = v[c] * dt * jv[c] * (r0 + rl) * 0.5f / dx;
        else             Vv = v[c] * dt * (r0 + rl) * 0.5f * jv[c] / dx;
    }
    if (k < d.nz - 1) {
        const float ru = RHO ? rho[c + d.sk] : 1.0f;
        Wv = w[c] * dt * jw[c] * (ru + r0) * 0.5f;
    } else
        Wv = w[c] * dt * jw[c] * r0;
    U[c] = Uv; V[c] = Vv; W[c] = Wv; Wdz[c] = Wv / dz[c];
}

// ------------------------------------------------------------------------------------------------
// A2: donor-cell pass with winds shared by all scalars (advect.f90:139-175, adv_mpdata.f90:44-105)
// ------------------------------------------------------------------------------------------------
template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_upwind_pass(Dims d, CVarPtrs in, VarPtrs out, int nv,
              const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ W,
              const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz,
              unsigned char *__restrict__ occ)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + threadIdx.y;
    const int j = tb.z;
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    // occ[m][j][k][it] = 1 when the 64-cell row segment (it,k,j) of the OUTPUT of scalar m holds a non-zero (pre-cleared
    // to 0; only ones are ever stored, so the two branches below cannot race)
    const size_t oslot = ((size_t)j * d.nz + k) * gridDim.x + tb.x, ostride = (size_t)d.ny * d.nz * gridDim.x;
    const bool interior = (i > 0) && (i < d.nx - 1) && (j > 0) && (j < d.ny - 1);
    if (!interior) {
        for (int m = 0; m < nv; ++m) {
            const float v = in.p[m][c];
            out.p[m][c] = v;
            if (occ && v != 0.0f) occ[(size_t)m * ostride + oslot] = 1;
        }
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
        float qq = q0 - fdiv((f1r - f1l) + (f3 - f4), den_h);
        if (bottom) {
            qq = qq - fdiv(flux1(q0, q[c + d.sk], Wt), den_v);
        } else if (top) {
            qq = qq - fdiv(q0 * Wt - flux1(q[c - d.sk], q0, Wb), den_v);
        } else {
            qq = qq - fdiv(flux1(q0, q[c + d.sk], Wt) - flux1(q[c - d.sk], q0, Wb), den_v);
        }
        out.p[m][c] = qq;
        if (occ) {                                               // one store per wave, by its first lane holding a non-zero
            const unsigned long long nzb = __ballot(qq != 0.0f);
            if (nzb && (int)threadIdx.x == __ffsll((long long)nzb) - 1) occ[(size_t)m * ostride + oslot] = 1;
        }
    }
}

// needf[m][block of k_mpdata_final2] = any non-zero of the pass-1 field within 2 cells of the block's outputs: outside of
// that, the unlimited velocity of every face the block touches is zero, fct_limit returns it unchanged and the final
// donor-cell pass reproduces the (zero) pass-1 field.
__global__ void k_occ_blocks(const unsigned char *__restrict__ occ, unsigned char *__restrict__ needf, int nt, int nx, int nz, int ny, int nv,
                             int gx, int gy, int gz, int fby, int fzs, int fjb, int xout)
{
    // one wave per (scalar, block): the lanes stride over the (j, k, i-segment) entries of the block's neighbourhood
    const size_t t = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    const size_t nb = (size_t)gx * gy * gz;
    if (t >= nb * nv) return;
    const int m = (int)(t / nb); const size_t r = t % nb;
    const int bx = (int)(r % gx), by = (int)((r / gx) % gy), bz = (int)(r / ((size_t)gx * gy));
    const int i0 = max(1 + bx * xout - 2, 0) / 64, i1 = min(1 + bx * xout + xout + 2, nx - 1) / 64;
    const int k0 = max(by * fzs - 2, 0), k1 = min(by * fzs + fby - 1 + 2, nz - 1);
    const int j0 = max(1 + bz * fjb - 2, 0), j1 = min(1 + bz * fjb + fjb - 1 + 2, ny - 1);
    const unsigned char *o = occ + (size_t)m * nt * nz * ny;
    const int ni = i1 - i0 + 1, nk = k1 - k0 + 1, tot = ni * nk * (j1 - j0 + 1);
    unsigned char any = 0;
    for (int e = lane; e < tot; e += 64) {
        const int ii = i0 + e % ni, kk = k0 + (e / ni) % nk, jj = j0 + e / (ni * nk);
        any |= o[((size_t)jj * nz + kk) * nt + ii];
    }
    const bool w = __any(any != 0);
    if (lane == 0) needf[t] = w ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// A3: anti-diffusive pseudo-velocities (adv_mpdata.f90:107-255) followed by the 0.5 / 0.5*dz
// scaling of advect3d (:383-385).  One thread = the three faces "owned" by cell (i,k,j):
// u2 on (i-1|i), v2 on (j-1|j), w2 above level k.
// ------------------------------------------------------------------------------------------------
template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_mpdata_fluxes(Dims d, CVarPtrs qin, VarPtrs u2o, VarPtrs v2o, VarPtrs w2o, int nv,
                const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ Wz,
                const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz,
                const unsigned char *__restrict__ occ)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + threadIdx.y;
    const int j = tb.z;
    // (all 64 lanes take part: this runs before the lanes beyond nx leave)
    // Which scalars have a non-zero anywhere in this row segment's stencil (the 27 neighbouring segments)?  The lanes
    // share the 27 x nv flag bytes between them, one ballot per scalar: a handful of loads per wave, no extra pass.
    unsigned needmask = ~0u;
    if (occ) {
        const int nt = (int)gridDim.x;
        const size_t ostride = (size_t)d.ny * d.nz * nt;
        needmask = 0;
        const int lane = threadIdx.x;
        for (int e = lane; e < 27 * nv; e += 64) {                 // e -> (scalar, neighbour)
            const int m = e / 27, nb = e - m * 27;
            const int jj = min(max(j + nb / 9 - 1, 0), d.ny - 1), kk = min(max(k + (nb / 3) % 3 - 1, 0), d.nz - 1);
            const int ii = min(max(tb.x + nb % 3 - 1, 0), nt - 1);
            if (occ[(size_t)m * ostride + ((size_t)jj * d.nz + kk) * nt + ii]) needmask |= 1u << m;
        }
        for (int dd = 32; dd > 0; dd >>= 1) needmask |= __shfl_xor(needmask, dd);
        needmask = __builtin_amdgcn_readfirstlane(needmask);
    }
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    const int sk = d.sk, sj = d.sj;
    const bool has_u = (i >= 1), has_v = (j >= 1), has_w = (k < d.nz - 1);
    const bool j_in = (j > 0) && (j < d.ny - 1);
    const bool k_in = (k > 0) && (k < d.nz - 1);
    const bool i_in = (i > 0) && (i < d.nx - 1);

    const float G0 = RHO ? jaco[c] * rho[c] : jaco[c];
    // scalar-independent pieces (hoisted out of the per-scalar loop)
    float u = 0, Gsu = 1, au = 0, cu_v = 0, cu_w = 0;
    if (has_u) {
        u = U[c];
        Gsu = G0 + (RHO ? jaco[c - 1] * rho[c - 1] : jaco[c - 1]);
        au = fabsf(u) * (1 - fabsf(u) / (0.5f * Gsu));
        if (j_in) cu_v = 0.5f * u * ((1 / 4.0f) * (V[c] + V[c + sj] + V[c - 1] + V[c - 1 + sj]));
        if (k_in) cu_w = 0.5f * u * ((1 / 4.0f) * (Wz[c] + Wz[c - sk] + Wz[c - 1] + Wz[c - 1 - sk]));
    }
    float v = 0, Gsv = 1, av = 0, cv_u = 0, cv_w = 0;
    if (has_v) {
        v = V[c];
        Gsv = G0 + (RHO ? jaco[c - sj] * rho[c - sj] : jaco[c - sj]);
        av = fabsf(v) * (1 - fabsf(v) / (0.5f * Gsv));
        // edge_v is zero at the x edges (adv_mpdata.f90:186-195), so the term is -0.5*v*0*0/G = -0
        const float ev = i_in ? (1 / 4.0f) * (U[c + 1] + U[c + 1 - sj] + U[c] + U[c - sj]) : 0.0f;
        cv_u = 0.5f * v * ev;
        if (k_in) cv_w = 0.5f * v * ((1 / 4.0f) * (Wz[c] + Wz[c - sk] + Wz[c - sj] + Wz[c - sk - sj]));
    }
    float w = 0, Gsw = 1, aw = 0, cw_u = 0, cw_v = 0, dzc = 0;
    if (has_w) {
        w = Wz[c];
        Gsw = (RHO ? jaco[c + sk] * rho[c + sk] : jaco[c + sk]) + G0;
        aw = fabsf(w) * (1 - fabsf(w) / (0.5f * Gsw));
        const float ev = i_in ? (1 / 4.0f) * (U[c + 1] + U[c + 1 + sk] + U[c] + U[c + sk]) : 0.0f;
        cw_u = 0.5f * w * ev;
        if (j_in) cw_v = 0.5f * w * ((1 / 4.0f) * (V[c] + V[c + sk] + V[c + sj] + V[c + sk + sj]));
        dzc = dz[c];
    }

    for (int m = 0; m < nv; ++m) {
        const float *__restrict__ q = qin.p[m];
        float *__restrict__ u2m = u2o.p[m], *__restrict__ v2m = v2o.p[m], *__restrict__ w2m = w2o.p[m];
        if (!((needmask >> m) & 1u)) {                            // wave-uniform: the whole stencil of this row segment is zero
            u2m[c] = 0.0f; v2m[c] = 0.0f; w2m[c] = 0.0f;
            continue;
        }
        const float q0 = q[c];
        // ---- U face (i-1 | i) : adv_mpdata.f90:134-168
        float r_u2 = 0.0f;
        if (has_u) {
            const float lx = q[c - 1];
            float val = fdiv(au * (q0 - lx), q0 + lx + 1e-10f);
            if (j_in) {
                const float a = q[c + sj], b = q[c - sj], e = q[c - 1 + sj], f = q[c - 1 - sj];
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cu_v * eq, Gsu);
            }
            if (k_in) {
                const float a = q[c + sk], b = q[c - sk], e = q[c - 1 + sk], f = q[c - 1 - sk];
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cu_w * eq, Gsu);
            }
            r_u2 = val * 0.5f;
        }
        u2m[c] = r_u2;
        // ---- V face (j-1 | j) : :172-208
        float r_v2 = 0.0f;
        if (has_v) {
            const float l = q[c - sj];
            float val = fdiv(av * (q0 - l), q0 + l + 1e-10f);
            {
                float eq = 0.0f;
                if (i_in) {
                    const float a = q[c + 1 - sj], b = q[c - 1], e = q[c + 1], f = q[c - 1 - sj];
                    eq = fdiv(a - b + e - f, e + a + b + f + 1e-10f);
                }
                val = val - fdiv(cv_u * eq, Gsv);
            }
            if (k_in) {
                const float a = q[c + sk - sj], b = q[c - sk], e = q[c + sk], f = q[c - sk - sj];
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cv_w * eq, Gsv);
            }
            r_v2 = val * 0.5f;
        }
        v2m[c] = r_v2;
        // ---- W face (k | k+1) : :214-249
        float r_w2 = 0.0f;
        if (has_w) {
            const float r = q[c + sk];
            float val = fdiv(aw * (r - q0), r + q0 + 1e-10f);
            {
                float eq = 0.0f;
                if (i_in) {
                    const float a = q[c + 1 + sk], b = q[c - 1], e = q[c + 1], f = q[c - 1 + sk];
                    eq = fdiv(a - b + e - f, e + a + b + f + 1e-10f);
                }
                val = val - fdiv(cw_u * eq, Gsw);
            }
            if (j_in) {
                const float a = q[c + sk + sj], b = q[c - sj], e = q[c + sj], f = q[c + sk - sj];
                const float eq = fdiv(a - b + e - f, e + f + a + b + 1e-10f);
                val = val - fdiv(cw_v * eq, Gsw);
            }
            r_w2 = val * 0.5f * dzc;
        }
        w2m[c] = r_w2;
    }
}

// ------------------------------------------------------------------------------------------------
// A3, software-pipelined over the scalars.  The version above issues the 16 stencil loads of ONE scalar, waits, computes,
// stores, and only then touches the next scalar: per wave one 256-B row segment of new HBM data is in flight at a time,
// and with 8 waves per SIMD that caps the kernel at ~2 TB/s of traffic, far below what the arithmetic needs
// (profiles/micro: replacing every IEEE division by v_rcp*mul removed half the VALU instructions and bought 8 %).
// Here the stencil of the NEXT active scalar is loaded (unconditionally, offsets clamped at the domain edges) before the
// current one is evaluated, so two scalars' worth of loads overlap the ~270 VALU instructions of one evaluation.
// Arithmetic: identical expressions in identical order -- results are bit-identical to k_mpdata_fluxes.
// ------------------------------------------------------------------------------------------------
// element of a REAL(4) array at `uniform base + 32-bit byte offset`: selects the saddr form of global_load / global_store
// (no per-load 64-bit VALU address arithmetic).  A tile is far below 4 GiB.
__device__ __forceinline__ float ldb(const float *__restrict__ p, unsigned b) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + b); }
__device__ __forceinline__ void stb(float *__restrict__ p, unsigned b, float v) { *reinterpret_cast<float *>(reinterpret_cast<char *>(p) + b) = v; }

struct Stencil16 {
    float c0, xm, xp, yp, ym, xm_yp, xm_ym, xp_ym, zp, zm, xm_zp, xm_zm, xp_zp, zp_ym, zm_ym, zp_yp;
};
// y / z offsets (elements) are wave-uniform and fold into the scalar base pointer; the x offsets are per-lane BYTE offsets
struct StencilOff { int yp, ym, zp, zm; unsigned b0, bxm, bxp; };

__device__ __forceinline__ Stencil16 load_stencil(const float *__restrict__ q, const StencilOff &o)
{
    Stencil16 s;
    s.c0 = ldb(q, o.b0); s.xm = ldb(q, o.bxm); s.xp = ldb(q, o.bxp); s.yp = ldb(q + o.yp, o.b0); s.ym = ldb(q + o.ym, o.b0);
    s.xm_yp = ldb(q + o.yp, o.bxm); s.xm_ym = ldb(q + o.ym, o.bxm); s.xp_ym = ldb(q + o.ym, o.bxp);
    s.zp = ldb(q + o.zp, o.b0); s.zm = ldb(q + o.zm, o.b0); s.xm_zp = ldb(q + o.zp, o.bxm); s.xm_zm = ldb(q + o.zm, o.bxm);
    s.xp_zp = ldb(q + o.zp, o.bxp); s.zp_ym = ldb(q + (o.zp + o.ym), o.b0); s.zm_ym = ldb(q + (o.zm + o.ym), o.b0);
    s.zp_yp = ldb(q + (o.zp + o.yp), o.b0);
    return s;
}

template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_mpdata_fluxes_pipe(Dims d, CVarPtrs qin, VarPtrs u2o, VarPtrs v2o, VarPtrs w2o, int nv,
                     const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ Wz,
                     const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz,
                     const unsigned char *__restrict__ occ)
{
    const TileId tb = xcd_tile(4);
    const int i = tb.x * BX + threadIdx.x;
    const int k = tb.y * BY + __builtin_amdgcn_readfirstlane(threadIdx.y);     // BX == 64: a wave is one k
    const int j = tb.z;
    unsigned needmask = (nv >= 32) ? ~0u : ((1u << nv) - 1u);
    if (occ) {                                                      // see k_mpdata_fluxes
        const int nt = (int)gridDim.x;
        const size_t ostride = (size_t)d.ny * d.nz * nt;
        needmask = 0;
        const int lane = threadIdx.x;
        for (int e = lane; e < 27 * nv; e += 64) {
            const int m = e / 27, nb = e - m * 27;
            const int jj = min(max(j + nb / 9 - 1, 0), d.ny - 1), kk = min(max(k + (nb / 3) % 3 - 1, 0), d.nz - 1);
            const int ii = min(max(tb.x + nb % 3 - 1, 0), nt - 1);
            if (occ[(size_t)m * ostride + ((size_t)jj * d.nz + kk) * nt + ii]) needmask |= 1u << m;
        }
        for (int dd = 32; dd > 0; dd >>= 1) needmask |= __shfl_xor(needmask, dd);
        needmask = __builtin_amdgcn_readfirstlane(needmask);
    }
    if (i >= d.nx || k >= d.nz) return;
    const int c = d.idx(i, k, j);
    const int sk = d.sk, sj = d.sj;
    const bool has_u = (i >= 1), has_v = (j >= 1), has_w = (k < d.nz - 1);
    const bool j_in = (j > 0) && (j < d.ny - 1);
    const bool k_in = (k > 0) && (k < d.nz - 1);
    const bool i_in = (i > 0) && (i < d.nx - 1);
    StencilOff o;
    o.b0 = 4u * (unsigned)c; o.bxm = has_u ? o.b0 - 4u : o.b0; o.bxp = (i < d.nx - 1) ? o.b0 + 4u : o.b0;
    o.ym = has_v ? -sj : 0; o.yp = (j < d.ny - 1) ? sj : 0;
    o.zm = (k > 0) ? -sk : 0; o.zp = has_w ? sk : 0;

    const float G0 = RHO ? jaco[c] * rho[c] : jaco[c];
    float u = 0, Gsu = 1, au = 0, cu_v = 0, cu_w = 0;
    if (has_u) {
        u = U[c];
        Gsu = G0 + (RHO ? jaco[c - 1] * rho[c - 1] : jaco[c - 1]);
        au = fabsf(u) * (1 - fabsf(u) / (0.5f * Gsu));
        if (j_in) cu_v = 0.5f * u * ((1 / 4.0f) * (V[c] + V[c + sj] + V[c - 1] + V[c - 1 + sj]));
        if (k_in) cu_w = 0.5f * u * ((1 / 4.0f) * (Wz[c] + Wz[c - sk] + Wz[c - 1] + Wz[c - 1 - sk]));
    }
    float v = 0, Gsv = 1, av = 0, cv_u = 0, cv_w = 0;
    if (has_v) {
        v = V[c];
        Gsv = G0 + (RHO ? jaco[c - sj] * rho[c - sj] : jaco[c - sj]);
        av = fabsf(v) * (1 - fabsf(v) / (0.5f * Gsv));
        const float ev = i_in ? (1 / 4.0f) * (U[c + 1] + U[c + 1 - sj] + U[c] + U[c - sj]) : 0.0f;
        cv_u = 0.5f * v * ev;
        if (k_in) cv_w = 0.5f * v * ((1 / 4.0f) * (Wz[c] + Wz[c - sk] + Wz[c - sj] + Wz[c - sk - sj]));
    }
    float w = 0, Gsw = 1, aw = 0, cw_u = 0, cw_v = 0, dzc = 0;
    if (has_w) {
        w = Wz[c];
        Gsw = (RHO ? jaco[c + sk] * rho[c + sk] : jaco[c + sk]) + G0;
        aw = fabsf(w) * (1 - fabsf(w) / (0.5f * Gsw));
        const float ev = i_in ? (1 / 4.0f) * (U[c + 1] + U[c + 1 + sk] + U[c] + U[c + sk]) : 0.0f;
        cw_u = 0.5f * w * ev;
        if (j_in) cw_v = 0.5f * w * ((1 / 4.0f) * (V[c] + V[c + sk] + V[c + sj] + V[c + sk + sj]));
        dzc = dz[c];
    }

    const unsigned all = (nv >= 32) ? ~0u : ((1u << nv) - 1u);
    for (unsigned z = all & ~needmask; z; z &= z - 1) {             // wave-uniform: the whole stencil of this row segment is zero
        const int m = __builtin_ctz(z);
        u2o.p[m][c] = 0.0f; v2o.p[m][c] = 0.0f; w2o.p[m][c] = 0.0f;
    }
    unsigned active = needmask & all;
    if (!active) return;
    int m = __builtin_ctz(active); active &= active - 1;
    Stencil16 s = load_stencil(qin.p[m], o);
    for (;;) {
        const int mn = active ? __builtin_ctz(active) : -1;
        Stencil16 nx_ = s;
        if (mn >= 0) nx_ = load_stencil(qin.p[mn], o);            // in flight while scalar m is evaluated
        // ---- U face (i-1 | i) : adv_mpdata.f90:134-168
        float r_u2 = 0.0f;
        if (has_u) {
            float val = fdiv(au * (s.c0 - s.xm), s.c0 + s.xm + 1e-10f);
            if (j_in) {
                const float a = s.yp, b = s.ym, e = s.xm_yp, f = s.xm_ym;
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cu_v * eq, Gsu);
            }
            if (k_in) {
                const float a = s.zp, b = s.zm, e = s.xm_zp, f = s.xm_zm;
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cu_w * eq, Gsu);
            }
            r_u2 = val * 0.5f;
        }
        // ---- V face (j-1 | j) : :172-208
        float r_v2 = 0.0f;
        if (has_v) {
            float val = fdiv(av * (s.c0 - s.ym), s.c0 + s.ym + 1e-10f);
            {
                float eq = 0.0f;
                if (i_in) {
                    const float a = s.xp_ym, b = s.xm, e = s.xp, f = s.xm_ym;
                    eq = fdiv(a - b + e - f, e + a + b + f + 1e-10f);
                }
                val = val - fdiv(cv_u * eq, Gsv);
            }
            if (k_in) {
                const float a = s.zp_ym, b = s.zm, e = s.zp, f = s.zm_ym;
                const float eq = fdiv(a - b + e - f, a + b + e + f + 1e-10f);
                val = val - fdiv(cv_w * eq, Gsv);
            }
            r_v2 = val * 0.5f;
        }
        // ---- W face (k | k+1) : :214-249
        float r_w2 = 0.0f;
        if (has_w) {
            float val = fdiv(aw * (s.zp - s.c0), s.zp + s.c0 + 1e-10f);
            {
                float eq = 0.0f;
                if (i_in) {
                    const float a = s.xp_zp, b = s.xm, e = s.xp, f = s.xm_zp;
                    eq = fdiv(a - b + e - f, e + a + b + f + 1e-10f);
                }
                val = val - fdiv(cw_u * eq, Gsw);
            }
            if (j_in) {
                const float a = s.zp_yp, b = s.ym, e = s.yp, f = s.zp_ym;
                const float eq = fdiv(a - b + e - f, e + f + a + b + 1e-10f);
                val = val - fdiv(cw_v * eq, Gsw);
            }
            r_w2 = val * 0.5f * dzc;
        }
        stb(u2o.p[m], o.b0, r_u2); stb(v2o.p[m], o.b0, r_v2); stb(w2o.p[m], o.b0, r_w2);
        if (mn < 0) break;
        s = nx_; m = mn; active &= active - 1;
    }
}

// ------------------------------------------------------------------------------------------------
// A4 core: limited pseudo-velocity of ONE face between cells a and b=a+1 of a 1-D line
// (adv_mpdata_FCT_core.f90:47-116 with the loop-carried values recomputed from their definition).
//   qm1,q0,q1,q2 : field after pass 1 at cells a-1,a,b,b+1     lm1,l0,l1,l2 : field before pass 1
//   Um,U0,Up     : unlimited pseudo-velocities on faces (a-1|a),(a|b),(b|b+1)
//   first: a is the first cell of the line; last: b is the last cell; is_w: vertical line
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fct_limit(float qm1, float q0, float q1, float q2,
                                           float lm1, float l0, float l1, float l2,
                                           float Um, float U0, float Up, bool first, bool last, bool is_w)
{
    // adv_mpdata_FCT_core.f90:47-116 has one code path for U0 > 0 (beta_out of the left cell, beta_in of the right
    // cell) and its mirror image for U0 < 0 (beta_in left, beta_out right).  The mirror image is the same formula
    // applied to the negated fields and fluxes:  max(x) = -min(-x),  qmax - q = (-q) - min(-x),
    // fin = max(0,fm) - min(0,f0) = max(0,-f0) - min(0,-fm).  Negation is exact and a-b == (-b)-(-a) bit for bit,
    // so flipping the sign bits of the inputs when U0 < 0 and running the U0 > 0 path gives identical results
    // without a divergent branch (antidiffusive velocities change sign from cell to cell).
    if (!(U0 > 0.0f) && !(U0 < 0.0f)) return U0;
    const unsigned sm = (U0 > 0.0f) ? 0u : 0x80000000u;
#define SGN(x) __uint_as_float(__float_as_uint(x) ^ sm)
    const float f0 = SGN(flux1(q0, q1, U0));
    const float a0 = SGN(q0), a1 = SGN(q1), b0 = SGN(l0), b1 = SGN(l1);
    float qmin_i, fout_i;
    if (first) {
        qmin_i = fminf(fminf(a0, a1), fminf(b0, b1));
        fout_i = is_w ? fmaxf(0.f, f0) : 0.0f;
    } else {
        const float fm = SGN(flux1(qm1, q0, Um));
        qmin_i = fminf(fminf(fminf(SGN(qm1), a0), fminf(a1, SGN(lm1))), fminf(b0, b1));
        fout_i = fmaxf(0.f, f0) - fminf(0.f, fm);
    }
    float qmax_i2, fin_i2;
    if (!last) {
        const float fp = SGN(flux1(q1, q2, Up));
        qmax_i2 = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(SGN(q2), b0)), fmaxf(b1, SGN(l2)));
        fin_i2 = fmaxf(0.f, f0) - fminf(0.f, fp);
    } else {
        qmax_i2 = fmaxf(fmaxf(a0, a1), b0);
        fin_i2 = is_w ? (fmaxf(0.f, f0) - fminf(0.f, f0)) : 0.0f;
    }
#undef SGN
    const float beta_out_i = fdiv(a0 - qmin_i, fout_i + 1e-15f);
    const float beta_in_i2 = fdiv(qmax_i2 - a1, fin_i2 + 1e-15f);
    return fminf(fminf(1.f, beta_in_i2), beta_out_i) * U0;
}

// ------------------------------------------------------------------------------------------------
// A4+A2 fused, second generation: every limited face is computed ONCE.
//   x : lane computes its LEFT face, the right face comes from lane+1 by a wave shuffle; tiles
//       overlap by one cell (63 outputs per 64 lanes) so no lane needs a second evaluation;
//   z : a wave computes its BOTTOM face, the top face comes from the wave above through LDS; k-chunks
//       overlap by one level (7 outputs per 8 waves);
//   y : each thread marches FJB rows; the north face of row j is the south face of row j+1 and the
//       j-direction neighbourhoods (q1, l, v2) roll through registers.
// ------------------------------------------------------------------------------------------------
#ifndef FBY
#define FBY 8
#endif
#ifndef FJB
#define FJB 8
#endif
// F2_NOBAR 0 (default): a wave limits the bottom face of its level, the top face comes from the wave above through LDS
// (one barrier per row; k-chunks overlap by one level).
// F2_NOBAR 1 (A/B build, profiles/micro/build_variant.py): every wave limits BOTH vertical faces itself -- one more
// fct_limit per cell, but no LDS hand-over, no barrier and no overlap (FBY outputs per FBY waves).  Same VALU work per
// output within 3 %; measured 1.00 ms against 0.96 ms: the per-row barrier is NOT what holds this kernel back.
#ifndef F2_NOBAR
#define F2_NOBAR 0
#endif
// F2_XSHFL 1: the x neighbours of q, l and u2 come from the neighbouring lanes' registers (q0 / l0 were loaded two rows
// earlier as the j+2 values, ux0 is the row's own load) instead of eight more loads per row that mostly missed the L1.
// Lanes 0, 1 and 63 are halo lanes (they only supply values): 60 outputs per 64 lanes instead of 63.
// F2_XSHFL 0: 63 outputs per 64 lanes, every lane loads its own x neighbours.
// (A first shuffle version kept 63 outputs and let lanes 0, 1, 63 load their neighbours: bit-identical, 1.11 ms against
// 0.97 ms -- the divergent three-lane loads cost more than the eight full-wave loads they replaced.)
#ifndef F2_XSHFL
#define F2_XSHFL 1
#endif
// F2_ZLDS 1: the z neighbours of q, l and w2 (levels k-2, k-1, k+1) come from the neighbouring waves through LDS -- each wave
// publishes its own row-ahead values (q, l at j+1 are already in its rolling registers) before the row's barrier -- and
// only the waves at the bottom / top of a block's k-chunk load theirs.  Interior waves: 5 loads per row instead of 13.
// A/B build only: bit-identical but slower than the loads it replaces -- 0.95 ms with the full denominator cache (48 kB of
// LDS: 3 blocks per CU), 0.98 ms with half of it (32 kB, jaco re-read), against 0.90 ms; the kernel then needs 67 VGPRs and
// spills 11 SGPRs, and forcing 64 VGPRs spills 43 SGPRs.
#ifndef F2_ZLDS
#define F2_ZLDS 0
#endif
#define F2_XOUT (F2_XSHFL ? 60 : 63)
#define F2_XL (F2_XSHFL ? 2 : 0)
#define F2_GX(nx) (F2_XSHFL ? ((nx) - 2 + F2_XOUT - 1) / F2_XOUT : ((nx) - 1 + 62) / 63)
#define FZS (F2_NOBAR ? FBY : FBY - 1)
template <bool RHO, bool FCT>
__global__ void __launch_bounds__(64 * FBY)
k_mpdata_final2(Dims d, CVarPtrs qold, CVarPtrs q1in, CVarPtrs u2i, CVarPtrs v2i, CVarPtrs w2i, VarPtrs out, int nv,
                const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz,
                const unsigned char *__restrict__ needf, int fjb, int xrows)
{
#if !F2_NOBAR
    __shared__ float s_wb[2][FBY][64];       // double-buffered by row parity: one barrier per row instead of two
#endif
    // jaco*rho and dz*jaco*rho of the thread's own FJB cells: computed by the first active scalar, re-used by the others.
    // (Re-reading jaco / dz / rho per scalar missed in L2 every time -- a scalar's march lasts far longer than L2 keeps a
    // line -- and was 0.75 GB of the kernel's 4.0 GB of fetch.)  Private slots: no barrier needed.
#if F2_ZLDS      // 4 blocks per CU leave 40 kB each: with the 12 kB of s_z only dz*jaco*rho is cached, jaco (rho) is re-read
    __shared__ float s_den[FJB][FBY][64];
#else
    __shared__ float2 s_den[FJB][FBY][64];
#endif
#if F2_ZLDS
    __shared__ float s_z[2][3][FBY][64];     // [row parity][q, l, w2][wave][lane]
#endif
    bool den_ready = false;
    // blockDim.x == 64: a wave is one ty, so everything derived from k is wave-uniform -- tell the compiler (readfirstlane)
    // so that the level offsets fold into the scalar base pointers and the bottom/top branches are scalar branches.
    const int lane = threadIdx.x, ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const TileId tb = xcd_tile(xrows);
    const int i = 1 + tb.x * F2_XOUT + lane - F2_XL;
    const int k = tb.y * FZS + ty;
    const int j0 = 1 + tb.z * fjb;
    const int j1 = min(j0 + fjb - 1, d.ny - 2);
    const int sk = d.sk, sj = d.sj;
    const bool in_i = (i <= d.nx - 1), in_k = (k <= d.nz - 1);
    const bool wave_out = in_k && (F2_NOBAR || ty <= FBY - 2 || k == d.nz - 1);          // wave-uniform
    const bool do_out = wave_out && (lane >= F2_XL) && (lane <= F2_XL + F2_XOUT - 1) && (i <= d.nx - 2);
    const bool bottom = (k == 0), top = (k == d.nz - 1);
    const int ic = in_i ? max(i, 0) : d.nx - 1, kc = in_k ? k : d.nz - 1;    // clamped => all loads stay in bounds
    const int cb = d.idx(ic, kc, 0);
    // Addressing: every array is read at `uniform base pointer + per-lane 32-bit BYTE offset` (global_load saddr form),
    // so a row costs four VALU address updates instead of a 64-bit add per load.  Lane-dependent are only the x clamps.
    const bool xfirst = (ic - 1 == 0), xlast = (ic == d.nx - 1);
#if !F2_XSHFL
    const int dxm2 = xfirst ? -4 : -8, dxp = xlast ? 0 : 4, dul = xfirst ? 0 : -4;
#endif
    const bool zfirst = (kc - 1 <= 0), zlast = (kc == d.nz - 1);
    const int ozm1 = bottom ? 0 : -sk, ozm2 = (kc >= 2) ? -2 * sk : ozm1, ozp = zlast ? 0 : sk;   // scalar
    const int ozp2 = (kc + 2 <= d.nz - 1) ? 2 * sk : ozp;
    // the level offsets as (wrapping) byte offsets added to the lane offset: five VALU adds per row, but one scalar
    // base pointer per array instead of one per array and level (those did not fit the SGPR file)
    const unsigned zb_m2 = 4u * (unsigned)ozm2, zb_m1 = 4u * (unsigned)ozm1, zb_p = 4u * (unsigned)ozp;
#if F2_NOBAR
    const unsigned zb_p2 = 4u * (unsigned)ozp2;
    const bool ztop1 = (kc + 1 == d.nz - 1);
#else
    (void)ozp2;
#endif
#if !F2_NOBAR
    unsigned rowctr = 0;                                          // rows processed by this block (block-uniform)
#endif
    unsigned needmask = ~0u;
    if (needf) {
        const size_t nblk = (size_t)gridDim.x * gridDim.y * gridDim.z, blk = tb.x + (size_t)gridDim.x * (tb.y + (size_t)gridDim.y * tb.z);
        needmask = 0;
        for (int m = 0; m < nv; ++m) needmask |= (needf[(size_t)m * nblk + blk] ? 1u : 0u) << m;
        needmask = __builtin_amdgcn_readfirstlane(needmask);
    }
    for (int m = 0; m < nv; ++m) {
        if (!((needmask >> m) & 1u)) {                            // block-uniform: pass-1 field is zero within 2 cells
            if (do_out) for (int j = j0; j <= j1; ++j) out.p[m][cb + j * sj] = 0.0f;
            continue;
        }
        const float *__restrict__ q = q1in.p[m];
        const float *__restrict__ l = qold.p[m];
        const float *__restrict__ u2 = u2i.p[m];
        const float *__restrict__ v2 = v2i.p[m];
        const float *__restrict__ w2 = w2i.p[m];
        float *__restrict__ o = out.p[m];
        // rolling y neighbourhood around row j: cells j-2..j+2, faces (j-1|j),(j|j+1),(j+1|j+2)
        float qm2 = 0, qm1, q0, qp1, qp2 = 0, lm2 = 0, lm1, l0, lp1, lp2 = 0, vm1 = 0, v0, vp1, vp2 = 0;
        float VS = 0, VN = 0;
        unsigned bc = 4u * (unsigned)(cb + j0 * sj);
        {
            qm1 = ldb(q - sj, bc); q0 = ldb(q, bc); qp1 = ldb(q + sj, bc);
            lm1 = ldb(l - sj, bc); l0 = ldb(l, bc); lp1 = ldb(l + sj, bc);
            v0 = ldb(v2, bc); vp1 = ldb(v2 + sj, bc);
            if (j0 - 2 >= 0) { qm2 = ldb(q - 2 * sj, bc); lm2 = ldb(l - 2 * sj, bc); vm1 = ldb(v2 - sj, bc); }
            if (FCT) VS = fct_limit(qm2, qm1, q0, qp1, lm2, lm1, l0, lp1, vm1, v0, vp1, j0 - 1 == 0, false, false);
            else VS = v0;
        }
#if F2_ZLDS
        float wcur = ldb(w2, bc);                                // w2 of the own level, row j0
        { const int p0 = (int)(rowctr & 1u); s_z[p0][0][ty][lane] = q0; s_z[p0][1][ty][lane] = l0; s_z[p0][2][ty][lane] = wcur; }
        __syncthreads();
#endif
        for (int j = j0; j <= j1; ++j, bc += 4u * (unsigned)sj) {
            const bool lastn = (j + 1 == d.ny - 1);
            // ---- all loads of this row up front, unconditional (clamped offsets) so they overlap ----
            const int oj2 = lastn ? sj : 2 * sj;
            qp2 = ldb(q + oj2, bc); lp2 = ldb(l + oj2, bc); vp2 = ldb(v2 + oj2, bc);
#if !F2_XSHFL
            const unsigned bxm2 = bc + (unsigned)dxm2, bxp = bc + (unsigned)dxp, bul = bc + (unsigned)dul;
#endif
#if F2_XSHFL
            // values differ from the loaded ones only where fct_limit ignores them (first / last cell of a line)
            const float ux0 = ldb(u2, bc);
            const float qxm2 = __shfl_up(q0, 2), qxm1 = __shfl_up(q0, 1), qxp1 = __shfl_down(q0, 1);
            const float lxm2 = __shfl_up(l0, 2), lxm1 = __shfl_up(l0, 1), lxp1 = __shfl_down(l0, 1);
            const float uxm = __shfl_up(ux0, 1), uxp = __shfl_down(ux0, 1);
#else
            const float qxm2 = ldb(q, bxm2), qxm1 = ldb(q - 1, bc), qxp1 = ldb(q, bxp);
            const float lxm2 = ldb(l, bxm2), lxm1 = ldb(l - 1, bc), lxp1 = ldb(l, bxp);
            const float uxm = ldb(u2, bul), ux0 = ldb(u2, bc), uxp = ldb(u2, bxp);
#endif
            // z: face (k-1|k); cells k-2,k-1,k,k+1 ; faces stored at the lower cell
            const unsigned bzm2 = bc + zb_m2, bzm1 = bc + zb_m1, bzp = bc + zb_p;
#if F2_ZLDS
            const int pz = (int)(rowctr & 1u);
            const float wzp = wcur;
            const float wnext = ldb(w2 + sj, bc);                // row j+1 (<= ny-1), published below for the next row
            float qzm2, qzm1, qzp1, lzm2, lzm1, lzp1, wzm, wz0;
            if (ty >= 2) { qzm2 = s_z[pz][0][ty - 2][lane]; lzm2 = s_z[pz][1][ty - 2][lane]; wzm = s_z[pz][2][ty - 2][lane]; }
            else         { qzm2 = ldb(q, bzm2); lzm2 = ldb(l, bzm2); wzm = ldb(w2, bzm2); }
            if (ty >= 1) { qzm1 = s_z[pz][0][ty - 1][lane]; lzm1 = s_z[pz][1][ty - 1][lane]; wz0 = s_z[pz][2][ty - 1][lane]; }
            else         { qzm1 = ldb(q, bzm1); lzm1 = ldb(l, bzm1); wz0 = ldb(w2, bzm1); }
            if (ty <= FBY - 2) { qzp1 = s_z[pz][0][ty + 1][lane]; lzp1 = s_z[pz][1][ty + 1][lane]; }
            else               { qzp1 = ldb(q, bzp); lzp1 = ldb(l, bzp); }
#else
            const float qzm2 = ldb(q, bzm2), qzm1 = ldb(q, bzm1), qzp1 = ldb(q, bzp);
            const float lzm2 = ldb(l, bzm2), lzm1 = ldb(l, bzm1), lzp1 = ldb(l, bzp);
            const float wzm = ldb(w2, bzm2), wz0 = ldb(w2, bzm1), wzp = ldb(w2, bc);
#endif
#if F2_NOBAR
            const unsigned bzp2 = bc + zb_p2;
            const float qzp2 = ldb(q, bzp2), lzp2 = ldb(l, bzp2), wzpp = ldb(w2, bzp);
            float WT = 0;
#endif
            float UL = 0, WB = 0;
            if (FCT) {
                if (wave_out) {
                    VN = fct_limit(qm1, q0, qp1, qp2, lm1, l0, lp1, lp2, v0, vp1, vp2, false, lastn, false);
                    UL = fct_limit(qxm2, qxm1, q0, qxp1, lxm2, lxm1, l0, lxp1, uxm, ux0, uxp, xfirst, xlast, false);
                }
                if (!bottom) WB = fct_limit(qzm2, qzm1, q0, qzp1, lzm2, lzm1, l0, lzp1, wzm, wz0, wzp, zfirst, zlast, true);
#if F2_NOBAR
                // face (k|k+1) == the bottom face of level k+1: the same call one level up
                if (!top) WT = fct_limit(qzm1, q0, qzp1, qzp2, lzm1, l0, lzp1, lzp2, wz0, wzp, wzpp, bottom, ztop1, true);
#endif
            } else {
                VN = vp1; UL = ux0;
                if (!bottom) WB = wz0;
#if F2_NOBAR
                if (!top) WT = wzp;
#endif
            }
            const float UR = __shfl_down(UL, 1);
#if !F2_NOBAR
            const int pb = (int)((rowctr++) & 1u);     // the buffer written two rows ago is free: everyone passed a barrier since
            s_wb[pb][ty][lane] = WB;
#if F2_ZLDS
            s_z[pb ^ 1][0][ty][lane] = qp1; s_z[pb ^ 1][1][ty][lane] = lp1; s_z[pb ^ 1][2][ty][lane] = wnext;
#endif
            __syncthreads();
            const float WT = (top || ty == FBY - 1) ? 0.0f : s_wb[pb][ty + 1][lane];
#endif
            if (do_out) {
                float den_h, den_v;
#if F2_ZLDS
                {
                    const float r = RHO ? ldb(rho, bc) : 1.0f;
                    const float ja = ldb(jaco, bc);
                    den_h = ja * r;
                    if (!den_ready) { den_v = ldb(dz, bc) * ja * r; s_den[j - j0][ty][lane] = den_v; }
                    else den_v = s_den[j - j0][ty][lane];
                }
#else
                if (!den_ready) {
                    const float r = RHO ? ldb(rho, bc) : 1.0f;
                    const float ja = ldb(jaco, bc);
                    den_h = ja * r;
                    den_v = ldb(dz, bc) * ja * r;
                    s_den[j - j0][ty][lane] = make_float2(den_h, den_v);
                } else {
                    const float2 t = s_den[j - j0][ty][lane];
                    den_h = t.x; den_v = t.y;
                }
#endif
                const float f1r = flux1(q0, qxp1, UR);
                const float f1l = flux1(qxm1, q0, UL);
                const float f3 = flux1(q0, qp1, VN);
                const float f4 = flux1(qm1, q0, VS);
                float qq = q0 - fdiv((f1r - f1l) + (f3 - f4), den_h);
                if (bottom) qq = qq - fdiv(flux1(q0, qzp1, WT), den_v);
                else if (top) qq = qq - fdiv(q0 * WT - flux1(qzm1, q0, WB), den_v);
                else qq = qq - fdiv(flux1(q0, qzp1, WT) - flux1(qzm1, q0, WB), den_v);
                stb(o, bc, qq);
            }
            qm2 = qm1; qm1 = q0; q0 = qp1; qp1 = qp2; lm2 = lm1; lm1 = l0; l0 = lp1; lp1 = lp2;
            vm1 = v0; v0 = vp1; vp1 = vp2; VS = VN;
#if F2_ZLDS
            wcur = wnext;
#endif
        }
        den_ready = true;
    }
}

// boundary ring of the new field := field after pass 1 (== field before the step), adv_mpdata.f90:63-65
__global__ void __launch_bounds__(256)
k_copy_ring(Dims d, CVarPtrs in, VarPtrs out, int nv)
{
    // ring cells: j in {0, ny-1} (full rows) and i in {0, nx-1}
    const int nrow = 2 * d.nx * d.nz, ncol = 2 * d.nz * (d.ny - 2);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrow + ncol) return;
    int i, k, j;
    if (t < nrow) { i = t % d.nx; k = (t / d.nx) % d.nz; j = (t / (d.nx * d.nz)) ? d.ny - 1 : 0; }
    else { const int s = t - nrow; k = s % d.nz; j = 1 + (s / d.nz) % (d.ny - 2); i = (s / (d.nz * (d.ny - 2))) ? d.nx - 1 : 0; }
    const int c = d.idx(i, k, j);
    for (int m = 0; m < nv; ++m) out.p[m][c] = in.p[m][c];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static dim3 grid3(const Dims &d) { return dim3((d.nx + BX - 1) / BX, (d.nz + BY - 1) / BY, d.ny); }

static int ensure_adv_scratch(icar_hip_ctx *c, int batch)
{
    const size_t bytes = c->n3 * sizeof(float);
    if (!c->U) {
        HIPCHK(hipMalloc(&c->U, bytes)); HIPCHK(hipMalloc(&c->V, bytes));
        HIPCHK(hipMalloc(&c->W, bytes)); HIPCHK(hipMalloc(&c->Wdz, bytes));
    }
    if (batch > c->batch_cap) {
        if (c->q2) { hipFree(c->q2); hipFree(c->u2); hipFree(c->v2); hipFree(c->w2); }
        HIPCHK(hipMalloc(&c->q2, bytes * batch)); HIPCHK(hipMalloc(&c->u2, bytes * batch));
        HIPCHK(hipMalloc(&c->v2, bytes * batch)); HIPCHK(hipMalloc(&c->w2, bytes * batch));
        c->batch_cap = batch;
    }
    return 0;
}

int icar_advect_setup_winds(icar_hip_ctx *c, int scheme, float dt, float dx, int advect_density)
{
    if (scheme != ICAR_ADV_UPWIND && scheme != ICAR_ADV_MPDATA) { icar_set_error("setup_winds: bad scheme"); return 1; }
    if (ensure_adv_scratch(c, 0)) return 1;
    const float *u = icar_field_f(c, ICAR_F_U), *v = icar_field_f(c, ICAR_F_V), *w = icar_field_f(c, ICAR_F_W);
    const float *ju = icar_field_f(c, ICAR_F_JACOBIAN_U), *jv = icar_field_f(c, ICAR_F_JACOBIAN_V);
    const float *jw = icar_field_f(c, ICAR_F_JACOBIAN_W), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    const float *rho = advect_density ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    if (!u || !v || !w || !ju || !jv || !jw || !dz || (advect_density && !rho)) return 1;
    ScopedTimer t(c, "winds");
    dim3 g = grid3(c->d), b(BX, BY);
#define LAUNCH(S, R) hipLaunchKernelGGL((k_setup_winds<S, R>), g, b, 0, c->stream, c->d, u, v, w, rho, ju, jv, jw, dz, dt, dx, c->U, c->V, c->W, c->Wdz)
    if (scheme == 1) { if (advect_density) LAUNCH(1, true); else LAUNCH(1, false); }
    else             { if (advect_density) LAUNCH(2, true); else LAUNCH(2, false); }
#undef LAUNCH
    HIPCHK(hipGetLastError());
    c->winds_valid = true;
    return 0;
}

// fraction of row segments (fluxes) / blocks (final pass) of the last MPDATA call that were NOT skipped, per scalar slot
int icar_advect_occupancy(icar_hip_ctx *c, int n, float *frac_fluxes, float *frac_final)
{
    if (!c->occ) { icar_set_error("advect_occupancy: no MPDATA call with occupancy flags yet"); return 1; }
    const int nt = (c->d.nx + BX - 1) / BX, nz = c->d.nz, ny = c->d.ny;
    const size_t per1 = (size_t)nt * nz * ny;
    const size_t perf = (size_t)F2_GX(c->d.nx) * ((F2_NOBAR ? nz + FBY - 1 : nz - 1 + FBY - 2) / FZS) * ((ny - 2 + FJB - 1) / FJB);
    std::vector<unsigned char> h1(per1 * n), hf(perf * n);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(h1.data(), c->occ, h1.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hf.data(), c->needf, hf.size(), hipMemcpyDeviceToHost));
    for (int m = 0; m < n; ++m) {
        size_t a = 0, b = 0;
        const unsigned char *o = h1.data() + m * per1;
        for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int it = 0; it < nt; ++it) {
            bool any = false;                                   // same 27-neighbour test as k_mpdata_fluxes
            for (int jj = std::max(j - 1, 0); jj <= std::min(j + 1, ny - 1) && !any; ++jj)
                for (int kk = std::max(k - 1, 0); kk <= std::min(k + 1, nz - 1) && !any; ++kk)
                    for (int ii = std::max(it - 1, 0); ii <= std::min(it + 1, nt - 1); ++ii)
                        if (o[((size_t)jj * nz + kk) * nt + ii]) { any = true; break; }
            a += any;
        }
        for (size_t t = 0; t < perf; ++t) b += hf[m * perf + t] != 0;
        frac_fluxes[m] = (float)a / per1; frac_final[m] = (float)b / perf;
    }
    return 0;
}

int icar_advect_run(icar_hip_ctx *c, int scheme, int order, int fct, int advect_density, const int *fields, int n)
{
    if (!c->winds_valid) { icar_set_error("advect: call icar_hip_setup_winds first"); return 1; }
    if (n <= 0) return 0;
    if (n > ICAR_MAX_ADV) { icar_set_error("advect: too many fields"); return 1; }
    if (scheme == ICAR_ADV_UPWIND) order = 1;
    if (order < 1) { icar_set_error("advect: mpdata_order must be >= 1"); return 1; }
    if (ensure_adv_scratch(c, order > 1 ? n : 0)) return 1;
    const float *rho = advect_density ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    if (!jaco || !dz || (advect_density && !rho)) return 1;
    CVarPtrs q, q2c, u2c, v2c, w2c; VarPtrs alt, q2, u2, v2, w2;
    for (int m = 0; m < n; ++m) {
        const int f = fields[m];
        if (f < 0 || f >= ICAR_N_ADVECTABLE) { icar_set_error("advect: field id is not an advectable scalar"); return 1; }
        for (int mm = 0; mm < m; ++mm) if (fields[mm] == f) { icar_set_error("advect: duplicate field"); return 1; }
        float *p = icar_field_f(c, f);
        if (!p) return 1;
        if (!c->alt[f]) HIPCHK(hipMalloc(&c->alt[f], c->n3 * sizeof(float)));
        q.p[m] = p; alt.p[m] = c->alt[f];
        if (order > 1) {
            q2.p[m] = c->q2 + (size_t)m * c->n3; u2.p[m] = c->u2 + (size_t)m * c->n3;
            v2.p[m] = c->v2 + (size_t)m * c->n3; w2.p[m] = c->w2 + (size_t)m * c->n3;
            q2c.p[m] = q2.p[m]; u2c.p[m] = u2.p[m]; v2c.p[m] = v2.p[m]; w2c.p[m] = w2.p[m];
        }
    }
    ScopedTimer t(c, "advect");
    dim3 g = grid3(c->d), b(BX, BY);
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
        if (advect_density) hipLaunchKernelGGL((k_upwind_pass<true>), g, b, 0, c->stream, c->d, q, alt, n, c->U, c->V, c->W, rho, jaco, dz, (unsigned char *)nullptr);
        else                hipLaunchKernelGGL((k_upwind_pass<false>), g, b, 0, c->stream, c->d, q, alt, n, c->U, c->V, c->W, rho, jaco, dz, (unsigned char *)nullptr);
        HIPCHK(hipGetLastError());
        swap_fields();
        return 0;
    }
    // Occupancy of the pass-1 fields: hydrometeor fields are zero over large parts of the domain, and a row segment
    // (fluxes) / block (final pass) whose whole stencil is zero produces exact zeros -- skipped, wave/block-uniformly.
    static const int fjb = getenv("ICAR_HIP_MPDATA_FJB") ? min(FJB, max(1, atoi(getenv("ICAR_HIP_MPDATA_FJB")))) : FJB;   // rows marched per block (<= FJB: s_den)
    static const int xrows = getenv("ICAR_HIP_MPDATA_XROWS") ? max(1, atoi(getenv("ICAR_HIP_MPDATA_XROWS"))) : 2;   // j slabs per XCD turn
    const dim3 gf(F2_GX(c->d.nx), (F2_NOBAR ? c->d.nz + FBY - 1 : c->d.nz - 1 + FBY - 2) / FZS, (c->d.ny - 2 + fjb - 1) / fjb), bf(64, FBY);
    const int nt = (int)g.x;
    const size_t occ_n = (size_t)ICAR_MAX_ADV * nt * c->d.nz * c->d.ny, nf_n = (size_t)ICAR_MAX_ADV * gf.x * gf.y * gf.z;
    static const bool no_skip = getenv("ICAR_HIP_MPDATA_NO_SKIP") != nullptr;      // A/B switch for profiling
    if (!c->occ && !no_skip) {
        HIPCHK(hipMalloc(&c->occ, occ_n)); HIPCHK(hipMalloc(&c->needf, nf_n));
    }
    unsigned char *occ = no_skip ? nullptr : c->occ;
    if (occ) HIPCHK(hipMemsetAsync(occ, 0, (size_t)n * nt * c->d.nz * c->d.ny, c->stream));
    // iord = 1 : q -> q2
    if (advect_density) hipLaunchKernelGGL((k_upwind_pass<true>), g, b, 0, c->stream, c->d, q, q2, n, c->U, c->V, c->W, rho, jaco, dz, occ);
    else                hipLaunchKernelGGL((k_upwind_pass<false>), g, b, 0, c->stream, c->d, q, q2, n, c->U, c->V, c->W, rho, jaco, dz, occ);
    if (occ) {
        const size_t n2 = (size_t)n * gf.x * gf.y * gf.z;
        hipLaunchKernelGGL(k_occ_blocks, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, c->stream, occ, c->needf, nt, c->d.nx, c->d.nz, c->d.ny, n,
                           (int)gf.x, (int)gf.y, (int)gf.z, FBY, FZS, fjb, F2_XOUT);
    }
    for (int iord = 2; iord <= order; ++iord) {
        // the flags describe the pass-1 field of the first corrective iteration only
        const unsigned char *need1 = (occ && iord == 2) ? occ : nullptr, *needf = (occ && iord == 2) ? c->needf : nullptr;
        // pseudo-velocities from q2 with the ORIGINAL U_m,V_m,W_m/dz (adv_mpdata.f90:379)
        static const bool pipe = getenv("ICAR_HIP_MPDATA_FLUXES") ? strcmp(getenv("ICAR_HIP_MPDATA_FLUXES"), "plain") != 0 : true;   // A/B switch
        if (pipe) {
            if (advect_density) hipLaunchKernelGGL((k_mpdata_fluxes_pipe<true>), g, b, 0, c->stream, c->d, q2c, u2, v2, w2, n, c->U, c->V, c->Wdz, rho, jaco, dz, need1);
            else                hipLaunchKernelGGL((k_mpdata_fluxes_pipe<false>), g, b, 0, c->stream, c->d, q2c, u2, v2, w2, n, c->U, c->V, c->Wdz, rho, jaco, dz, need1);
        } else {
            if (advect_density) hipLaunchKernelGGL((k_mpdata_fluxes<true>), g, b, 0, c->stream, c->d, q2c, u2, v2, w2, n, c->U, c->V, c->Wdz, rho, jaco, dz, need1);
            else                hipLaunchKernelGGL((k_mpdata_fluxes<false>), g, b, 0, c->stream, c->d, q2c, u2, v2, w2, n, c->U, c->V, c->Wdz, rho, jaco, dz, need1);
        }
        // limiter (l = q, q1 = q2) fused into the donor-cell pass q2 -> alt ; then q := alt
        {
            const int nring = 2 * c->d.nx * c->d.nz + 2 * c->d.nz * (c->d.ny - 2);
            hipLaunchKernelGGL(k_copy_ring, dim3((nring + 255) / 256), dim3(256), 0, c->stream, c->d, q2c, alt, n);
        }
#define FINAL(R, F) hipLaunchKernelGGL((k_mpdata_final2<R, F>), gf, bf, 0, c->stream, c->d, q, q2c, u2c, v2c, w2c, alt, n, rho, jaco, dz, needf, fjb, xrows)
        if (advect_density) { if (fct) FINAL(true, true); else FINAL(true, false); }
        else                { if (fct) FINAL(false, true); else FINAL(false, false); }
#undef FINAL
        swap_fields();
        if (iord != order) {
            // adv_mpdata.f90:393-402 : q2 := q before the next corrective iteration
            for (int m = 0; m < n; ++m)
                HIPCHK(hipMemcpyAsync(q2.p[m], q.p[m], c->n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}
