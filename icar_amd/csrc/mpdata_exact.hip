// icar_amd/csrc/mpdata_exact.hip -- MPDATA in the REFERENCE'S OWN OPERATION ORDER (icar_hip_mpdata_exact(ctx, 1)): every sum,
// product and quotient of adv_mpdata.f90 evaluated as written (IEEE division, no contraction, the C library's tie rule for
// max / min), so that the advected fields are bit-identical to the CPU reference and a whole sub-step sequence (microphysics ->
// halo -> advection -> forcing) can be compared bit for bit over many steps.  The fused kernel of mpdata.hip stays the default:
// it reads and writes each scalar once; this path keeps the reference's stages as launches (donor cell of all scalars, velocities of
// all scalars, limiter + donor cell per scalar) and exchanges q2, u2, v2, w2 through HBM (~60 B per scalar-cell instead of 8).
//
//   donor cell   q  -> q2                      adv_mpdata.f90:44-105, :374      k_upwind_pass (advect.hip, bit-exact since round 1)
//   velocities   q2 -> u2, v2, w2 (* 0.5, dz)  adv_mpdata.f90:107-255, :383-385 k_mpx_velocities
//   limiter      q, q2, u2.. -> limited u2..   adv_mpdata.f90:257-354 + adv_mpdata_FCT_core.f90:47-116  } k_mpx_limit_donor
//   donor cell   q2 -> q with the limited pseudo-velocities   :389                                         } (one launch)
//   mpdata_order > 2: q2 := q and again from "velocities" with the original U_m, V_m, W_m/dz (:379, :393-402)
//
// The limiter's carried variables (qmax_i, qmax_i2, fin_i, ...: adv_mpdata_FCT_core.f90 walks a line and hands "i2" of one face
// to "i" of the next) are functions of a face's three nearest unlimited fluxes and the four cells around it, so faces are
// independent, and always formed from the UNLIMITED velocities of the neighbouring faces (the f(:) of flux1 is computed before the
// line is walked, :47).
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
k_mpx_velocities(Dims d, int nv, const float *__restrict__ q0, size_t qstride, const float *__restrict__ u, const float *__restrict__ v,
                 const float *__restrict__ w, const float *__restrict__ rho, const float *__restrict__ jaco,
                 const float *__restrict__ dz, float *__restrict__ u2, float *__restrict__ v2, float *__restrict__ w2)
{
    // scalar m: field q0 + m * qstride, results u2 / v2 / w2 + m * qstride.  Everything that does not depend on the scalar -- the
    // leading factor |U| (1 - |U| / (0.5 (G + G'))), the sum G + G' and the products 0.5 U Ubar_perp of the cross terms, each in
    // the reference's own order of operations -- is evaluated once per thread and reused for the nv scalars.
    const int i = blockIdx.x * BX + threadIdx.x;
    const int k = blockIdx.y * BY + threadIdx.y;
    const int j = blockIdx.z;
    if (i >= d.nx || k >= d.nz) return;
    const int nx = d.nx, nz = d.nz, ny = d.ny;
    const int c = d.idx(i, k, j);
#define G(cc) (jaco[cc] * (RHO ? rho[cc] : 1.0f))
    const bool fu = i >= 1, fv = j >= 1, fw = k < nz - 1;
    const bool jin = (j > 0) && (j < ny - 1), kin = (k > 0) && (k < nz - 1), iin = (i > 0) && (i < nx - 1);
    const float Gc = G(c), dzc = dz[c];
    float aU = 0, GsU = 1, cUV = 0, cUW = 0, aV = 0, GsV = 1, cVU = 0, cVW = 0, aW = 0, GsW = 1, cWU = 0, cWV = 0;
    if (fu) {                                                                                   // :134-169
        GsU = Gc + G(c - 1);
        aU = fabsf(u[c]) * (1 - fabsf(u[c]) / (0.5f * GsU));
        if (jin) cUV = 0.5f * u[c] * ((1 / 4.0f) * (v[c] + v[c + d.sj] + v[c - 1] + v[c - 1 + d.sj]));
        if (kin) cUW = 0.5f * u[c] * ((1 / 4.0f) * (w[c] + w[c - d.sk] + w[c - 1] + w[c - 1 - d.sk]));
    }
    if (fv) {                                                                                   // :172-208
        GsV = Gc + G(c - d.sj);
        aV = fabsf(v[c]) * (1 - fabsf(v[c]) / (0.5f * GsV));
        float ev = 0;
        if (iin) ev = (1 / 4.0f) * (u[c + 1] + u[c + 1 - d.sj] + u[c] + u[c - d.sj]);
        cVU = 0.5f * v[c] * ev;
        if (kin) cVW = 0.5f * v[c] * ((1 / 4.0f) * (w[c] + w[c - d.sk] + w[c - d.sj] + w[c - d.sk - d.sj]));
    }
    if (fw) {                                                                                   // :214-249
        GsW = G(c + d.sk) + Gc;
        aW = fabsf(w[c]) * (1 - fabsf(w[c]) / (0.5f * GsW));
        float ev = 0;
        if (iin) ev = (1 / 4.0f) * (u[c + 1] + u[c + 1 + d.sk] + u[c] + u[c + d.sk]);
        cWU = 0.5f * w[c] * ev;
        if (jin) cWV = 0.5f * w[c] * ((1 / 4.0f) * (v[c] + v[c + d.sk] + v[c + d.sj] + v[c + d.sk + d.sj]));
    }
#undef G
    for (int m = 0; m < nv; ++m) {
        const float *__restrict__ q = q0 + (size_t)m * qstride;
#define Q(ii, kk, jj) q[d.idx(ii, kk, jj)]
        float ru = 0.0f, rv = 0.0f, rw = 0.0f;
        if (fu) {
            const float rx = Q(i, k, j), lx = Q(i - 1, k, j);
            const float denomx = (rx + lx + 1e-10f);
            float val = aU * (rx - lx) / denomx;
            if (jin) {                                                                          // UxV
                const float eq = (Q(i, k, j + 1) - Q(i, k, j - 1) + Q(i - 1, k, j + 1) - Q(i - 1, k, j - 1)) /
                                 (Q(i, k, j + 1) + Q(i, k, j - 1) + Q(i - 1, k, j + 1) + Q(i - 1, k, j - 1) + 1e-10f);
                val = val - cUV * eq / GsU;
            }
            if (kin) {                                                                          // UxW
                const float eq = (Q(i, k + 1, j) - Q(i, k - 1, j) + Q(i - 1, k + 1, j) - Q(i - 1, k - 1, j)) /
                                 (Q(i, k + 1, j) + Q(i, k - 1, j) + Q(i - 1, k + 1, j) + Q(i - 1, k - 1, j) + 1e-10f);
                val = val - cUW * eq / GsU;
            }
            ru = val;
        }
        if (fv) {
            const float r = Q(i, k, j), l = Q(i, k, j - 1);
            const float denom = (r + l + 1e-10f);
            float val = aV * (r - l) / denom;
            {                                                                                   // VxU (zero in the x ring)
                float eq = 0;
                if (iin)
                    eq = (Q(i + 1, k, j - 1) - Q(i - 1, k, j) + Q(i + 1, k, j) - Q(i - 1, k, j - 1)) /
                         (Q(i + 1, k, j) + Q(i + 1, k, j - 1) + Q(i - 1, k, j) + Q(i - 1, k, j - 1) + 1e-10f);
                val = val - cVU * eq / GsV;
            }
            if (kin) {                                                                          // VxW
                const float eq = (Q(i, k + 1, j - 1) - Q(i, k - 1, j) + Q(i, k + 1, j) - Q(i, k - 1, j - 1)) /
                                 (Q(i, k + 1, j - 1) + Q(i, k - 1, j) + Q(i, k + 1, j) + Q(i, k - 1, j - 1) + 1e-10f);
                val = val - cVW * eq / GsV;
            }
            rv = val;
        }
        if (fw) {
            const float r = Q(i, k + 1, j), l = Q(i, k, j);
            const float denom = (r + l + 1e-10f);
            float val = aW * (r - l) / denom;
            {                                                                                   // WxU
                float eq = 0;
                if (iin)
                    eq = (Q(i + 1, k + 1, j) - Q(i - 1, k, j) + Q(i + 1, k, j) - Q(i - 1, k + 1, j)) /
                         (Q(i + 1, k, j) + Q(i + 1, k + 1, j) + Q(i - 1, k, j) + Q(i - 1, k + 1, j) + 1e-10f);
                val = val - cWU * eq / GsW;
            }
            if (jin) {                                                                          // WxV
                const float eq = (Q(i, k + 1, j + 1) - Q(i, k, j - 1) + Q(i, k, j + 1) - Q(i, k + 1, j - 1)) /
                                 (Q(i, k, j + 1) + Q(i, k + 1, j - 1) + Q(i, k + 1, j + 1) + Q(i, k, j - 1) + 1e-10f);
                val = val - cWV * eq / GsW;
            }
            rw = val;
        }
#undef Q
        u2[(size_t)m * qstride + c] = ru * 0.5f;                                                 // :383-385
        v2[(size_t)m * qstride + c] = rv * 0.5f;
        w2[(size_t)m * qstride + c] = rw * 0.5f * dzc;
    }
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
    // :96-116.  U > 0: min(min(1, beta_in_i2), beta_out_i) U; U < 0: min(min(1, beta_in_i), beta_out_i2) U; U == 0 stays.  The two
    // quotients a face needs are selected BEFORE they are formed (lanes of a wave disagree about the sign: both branches would
    // cost four IEEE divisions)
    const bool pos = Uc > 0;
    const float num_i  = pos ? (q1c - qmin_i) : (qmax_i - q1c),    den_i  = pos ? fout_i : fin_i;       // beta_out_i  | beta_in_i
    const float num_i2 = pos ? (qmax_i2 - q1p) : (q1p - qmin_i2),  den_i2 = pos ? fin_i2 : fout_i2;     // beta_in_i2  | beta_out_i2
    const float beta_i = num_i / (den_i + 1e-15f), beta_i2 = num_i2 / (den_i2 + 1e-15f);
    const float first = pos ? beta_i2 : beta_i, second = pos ? beta_i : beta_i2;
    const float U = (pos || Uc < 0) ? o_min(o_min(1.f, first), second) * Uc : Uc;
    return U;
}

// flux_limiter (adv_mpdata.f90:257-354): x lines and z lines of the rows j = 2 .. ny-1 (z lines of the columns i = 2 .. nx-1 only),
// y lines of every (i, k).  l = the field the iteration started from, q1 = after the donor-cell pass.
// limited velocity of the x face between cells i-1 and i (1 <= i <= nx-1) of an interior row: t = i-1 of a line of nx cells
__device__ __forceinline__ float lim_u(const Dims &d, const float *__restrict__ l, const float *__restrict__ q1, const float *__restrict__ u2, int i, int c)
{
    const int t = i - 1;
    const bool hm = t > 0, hp = t != d.nx - 2;
    return fct_face(t, d.nx, false, hm ? q1[c - 2] : 0.f, q1[c - 1], q1[c], hp ? q1[c + 1] : 0.f,
                    hm ? l[c - 2] : 0.f, l[c - 1], l[c], hp ? l[c + 1] : 0.f,
                    hm ? u2[c - 1] : 0.f, u2[c], hp ? u2[c + 1] : 0.f);
}
// y face between j-1 and j (1 <= j <= ny-1): t = j-1 of a line of ny cells
__device__ __forceinline__ float lim_v(const Dims &d, const float *__restrict__ l, const float *__restrict__ q1, const float *__restrict__ v2, int j, int c)
{
    const int t = j - 1, s = d.sj;
    const bool hm = t > 0, hp = t != d.ny - 2;
    return fct_face(t, d.ny, false, hm ? q1[c - 2 * s] : 0.f, q1[c - s], q1[c], hp ? q1[c + s] : 0.f,
                    hm ? l[c - 2 * s] : 0.f, l[c - s], l[c], hp ? l[c + s] : 0.f,
                    hm ? v2[c - s] : 0.f, v2[c], hp ? v2[c + s] : 0.f);
}
// z face above level k (0 <= k <= nz-2) of an interior column: t = k of a line of nz cells; w(kme) = 0 afterwards (:322)
__device__ __forceinline__ float lim_w(const Dims &d, const float *__restrict__ l, const float *__restrict__ q1, const float *__restrict__ w2, int k, int c)
{
    const int t = k, s = d.sk;
    const bool hm = t > 0, hp = t != d.nz - 2;
    return fct_face(t, d.nz, true, hm ? q1[c - s] : 0.f, q1[c], q1[c + s], hp ? q1[c + 2 * s] : 0.f,
                    hm ? l[c - s] : 0.f, l[c], l[c + s], hp ? l[c + 2 * s] : 0.f,
                    hm ? w2[c - s] : 0.f, w2[c], hp ? w2[c + s] : 0.f);
}

// The limiter and the donor-cell pass that uses its result (:389), one launch.  An interior cell needs the limited velocities of
// its six faces; the three limited fields never go through HBM.  A face belongs to two cells: the x face is evaluated ONCE (by
// the cell on its right; the cell on its left takes it from the next lane -- x tiles overlap by one lane for that), the z face
// once (by the cell below it; the cell above takes it from LDS, the lowest wave of a block evaluates its own lower face), the
// y faces by both cells (the rows of a block's neighbours in y are other blocks): 4.25 face evaluations per cell instead of 6
// (181 -> 175 us per scalar at 512 x 512 x 40 -- less than the instruction count suggests; ~120 VALU instructions per face: three flux1, two extrema sets with the C
// library's tie rule, two IEEE divisions).  Then upwind_advection (adv_mpdata.f90:44-105, the expression of k_upwind_pass in
// advect.hip) with them.  Faces of boundary cells are not needed: the donor-cell pass updates interior cells only.
#define MPX_TX (BX - 1)                 // cells a block owns along x; lane BX-1 only provides its left face to lane BX-2
template <bool RHO>
__global__ void __launch_bounds__(BX * BY)
k_mpx_limit_donor(Dims d, const float *__restrict__ l, const float *__restrict__ q1,
                  const float *__restrict__ u2, const float *__restrict__ v2, const float *__restrict__ w2,
                  const float *__restrict__ rho, const float *__restrict__ jaco, const float *__restrict__ dz, float *__restrict__ out)
{
    __shared__ float s_w[BY][BX];
    const int lane = threadIdx.x, ty = threadIdx.y;
    const int i = blockIdx.x * MPX_TX + lane;
    const int k = blockIdx.y * BY + ty;
    const int j = blockIdx.z;
    const bool valid = (i < d.nx) && (k < d.nz);
    const int c = valid ? d.idx(i, k, j) : 0;
    const bool row = (j > 0) && (j < d.ny - 1), col = (i > 0) && (i < d.nx - 1);
    const bool interior = valid && row && col;
    const bool bottom = (k == 0), top = (k == d.nz - 1);
    // x: my left face (between i-1 and i), for every cell 1 <= i <= nx-1 of an interior row
    float Ul = 0.f;
    if (valid && row && i >= 1) Ul = lim_u(d, l, q1, u2, i, c);
    const float Ur = __shfl_down(Ul, 1);                                          // the left face of the cell to my right
    // z: the face above me
    float Wt = 0.f;
    if (interior && !top) Wt = lim_w(d, l, q1, w2, k, c);
    s_w[ty][lane] = Wt;
    __syncthreads();
    float Wb = 0.f;
    if (ty > 0) Wb = s_w[ty - 1][lane];
    else if (interior && !bottom) Wb = lim_w(d, l, q1, w2, k - 1, c - d.sk);      // (wave-uniform: ty is the wave)
    if (!valid || lane == BX - 1) return;
    const float q0 = q1[c];
    if (!interior) { out[c] = q0; return; }
    const float Vs = lim_v(d, l, q1, v2, j, c), Vn = lim_v(d, l, q1, v2, j + 1, c + d.sj);
    const float r = RHO ? rho[c] : 1.0f;
    const float ja = jaco[c];
    const float den_h = ja * r;
    const float den_v = dz[c] * ja * r;
    const float f1r = flux1(q0, q1[c + 1], Ur);
    const float f1l = flux1(q1[c - 1], q0, Ul);
    const float f3 = flux1(q0, q1[c + d.sj], Vn);
    const float f4 = flux1(q1[c - d.sj], q0, Vs);
    float qq = q0 - ((f1r - f1l) + (f3 - f4)) / den_h;
    if (bottom)   qq = qq - flux1(q0, q1[c + d.sk], Wt) / den_v;
    else if (top) qq = qq - (q0 * Wt - flux1(q1[c - d.sk], q0, Wb)) / den_v;
    else          qq = qq - (flux1(q0, q1[c + d.sk], Wt) - flux1(q1[c - d.sk], q0, Wb)) / den_v;
    out[c] = qq;
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
    // q2 of every scalar (two sets: mpdata_order > 2 ping-pongs) and u2, v2, w2 of every scalar
    const size_t n3 = c->n3, per = (size_t)nv * n3;
    if (c->mpx_nv < nv) {
        if (c->mpx_buf) { (void)hipFree(c->mpx_buf); c->mpx_buf = nullptr; c->mpx_nv = 0; }
        HIPCHK(hipMalloc(&c->mpx_buf, 5 * per * sizeof(float)));
        c->mpx_nv = nv;
    }
    float *cur = c->mpx_buf, *other = cur + per, *u2 = other + per, *v2 = u2 + per, *w2 = v2 + per;
    const dim3 g((d.nx + BX - 1) / BX, (d.nz + BY - 1) / BY, d.ny), b(BX, BY);
    const dim3 gl((d.nx + MPX_TX - 1) / MPX_TX, (d.nz + BY - 1) / BY, d.ny);           // limiter + donor cell: x tiles overlap by one lane
    VarPtrs out;
    for (int m = 0; m < nv; ++m) out.p[m] = cur + (size_t)m * n3;
    if (icar_upwind_pass_run(c, rho_on, q, out, nv, c->U, c->V, c->W)) return 1;                  // iord = 1 (:374), all scalars
    for (int iord = 2; iord <= order; ++iord) {
        if (rho_on) hipLaunchKernelGGL((k_mpx_velocities<true>), g, b, 0, c->stream, d, nv, cur, n3, c->U, c->V, c->Wdz, rho, jaco, dz, u2, v2, w2);
        else        hipLaunchKernelGGL((k_mpx_velocities<false>), g, b, 0, c->stream, d, nv, cur, n3, c->U, c->V, c->Wdz, rho, jaco, dz, u2, v2, w2);
        HIPCHK(hipGetLastError());
        for (int m = 0; m < nv; ++m) {
            const size_t o = (size_t)m * n3;
            const float *q2 = cur + o;
            const float *l = (iord == 2) ? q.p[m] : q2;                                           // :393-402: from iord = 3 on q == q2
            float *dst = (iord == order) ? alt.p[m] : other + o;
            if (fct) {                                                                            // limiter + :389
                if (rho_on) hipLaunchKernelGGL((k_mpx_limit_donor<true>), gl, b, 0, c->stream, d, l, q2, u2 + o, v2 + o, w2 + o, rho, jaco, dz, dst);
                else        hipLaunchKernelGGL((k_mpx_limit_donor<false>), gl, b, 0, c->stream, d, l, q2, u2 + o, v2 + o, w2 + o, rho, jaco, dz, dst);
                HIPCHK(hipGetLastError());
            } else {
                CVarPtrs in1; VarPtrs out1;
                in1.p[0] = q2; out1.p[0] = dst;
                if (icar_upwind_pass_run(c, rho_on, in1, out1, 1, u2 + o, v2 + o, w2 + o)) return 1;   // :389
            }
        }
        std::swap(cur, other);                                                                    // q2 := q (:393-402)
    }
    return 0;
}
