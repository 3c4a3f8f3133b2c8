// icar_amd/csrc/iterative_winds.hip -- SURVEY 8(f) rank 4: iterative_winds (src/physics/wind.f90:371-498)
// and the staggered halo boxes its exchange_u / exchange_v need (src/objects/exchangeable_obj.f90:158-229).
//
// The reference's loop body is three whole-array passes per iteration (calc_divergence, ADJ = div/ADJ_coef, four array
// statements on u and v) followed by exchange_u/exchange_v.  Here one iteration is two streaming kernels:
//   k_iw_adj    one thread per cell: div (wind.f90:203-226) and ADJ = div / (-2/dx) in registers, one store
//   k_iw_apply  one thread per cell: the u face i and the v face j of that cell, each `x + ADJ(lo)*0.5` then
//               `x - ADJ(hi)*0.5` (two roundings, as the two array statements per component do)
// A Jacobi sweep needs the whole ADJ field before any face moves, hence the kernel boundary.  HBM-bound: per cell
// and sweep 8 float reads + 1 write (adj) and 4 reads + 2 writes (apply) = 60 B.
#include "ctx.h"

namespace {

__global__ void k_iw_correct_w(Dims d, float *__restrict__ w, const float *__restrict__ dz)
{
    // wind.f90:430-441.  sum(dz(i,:,j)) and sum(dz(i,1:k,j)) accumulate bottom-up, so the partial sums are one running sum.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= d.nx) return;
    float height = 0.0f;
    for (int k = 0; k < d.nz; ++k) height += dz[d.idx(i, k, j)];
    const float wtop = w[d.idx(i, d.nz - 1, j)];
    float part = 0.0f;
    for (int k = 0; k < d.nz; ++k) {
        const int c = d.idx(i, k, j);
        part += dz[c];
        const float corr = fminf(part / height, 1.0f);
        w[c] = w[c] - corr * wtop;     // the top level reads its own (still uncorrected) value too: k ascends
    }
}

__global__ void k_iw_adj(Dims d, const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w,
                         const float *__restrict__ ju, const float *__restrict__ jv, const float *__restrict__ jw,
                         const float *__restrict__ dz, const float *__restrict__ jaco, float dx, float *__restrict__ adj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y, j = blockIdx.z;
    if (i >= d.nx) return;
    const int c = d.idx(i, k, j);
    const int cu = i + (d.nx + 1) * (k + d.nz * j);
    const float du = u[cu + 1] * ju[cu + 1] - u[cu] * ju[cu];           // :203-207
    const float dv = v[c + d.sj] * jv[c + d.sj] - v[c] * jv[c];         // :205-208
    float div = (du + dv) / dx;                                         // :210
    const float wm = w[c] * jw[c];                                      // :213
    if (k == 0) div = div + wm / dz[c];                                 // :217
    else        div = div + (wm - w[c - d.sk] * jw[c - d.sk]) / dz[c];  // :219
    div = div / jaco[c];                                                // :225
    const float coef = -2 / dx;                                         // ADJ_coef :444
    adj[c] = div / coef;                                                // :460
}

__global__ void k_iw_apply(Dims d, float *__restrict__ u, float *__restrict__ v, const float *__restrict__ adj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y, j = blockIdx.z;
    if (i >= d.nx) return;
    const int c = d.idx(i, k, j);
    const float a = adj[c] * 0.5f;
    if (i >= 2 && j >= 1 && j <= d.ny - 2) {                            // u(ims+2:ime, :, jms+1:jme-1)  :464-467
        const int cu = i + (d.nx + 1) * (k + d.nz * j);
        float x = u[cu];
        x = x + adj[c - 1] * 0.5f;
        x = x - a;
        u[cu] = x;
    }
    if (j >= 2 && i >= 1 && i <= d.nx - 2) {                            // v(ims+1:ime-1, :, jms+2:jme)  :473-476
        float x = v[c];
        x = x + adj[c - d.sj] * 0.5f;
        x = x - a;
        v[c] = x;
    }
}

// box <-> contiguous buffer [nj][nz][ni]
template <bool UNPACK>
__global__ void k_box(int X, int nz, int i0, int ni, int j0, int nj, float *__restrict__ f, float *__restrict__ buf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y, jj = blockIdx.z;
    if (t >= ni) return;
    const size_t a = (size_t)(i0 + t) + (size_t)X * (k + (size_t)nz * (j0 + jj));
    const size_t b = (size_t)t + (size_t)ni * (k + (size_t)nz * jj);
    if (UNPACK) f[a] = buf[b]; else buf[b] = f[a];
}

__global__ void k_divide(size_t n, float *__restrict__ x, const float *__restrict__ a)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) x[t] = x[t] / a[t];
}

struct Winds { float *u, *v, *w; };

int pick_winds(icar_hip_ctx *c, int update, Winds *o)
{
    if (update) {
        o->u = c->dqdt[ICAR_F_U]; o->v = c->dqdt[ICAR_F_V]; o->w = c->dqdt[ICAR_F_W];
        if (!o->u || !o->v || !o->w) { icar_set_error("iterative_winds(update): u/v dqdt_3d not uploaded or balance_uvw_update not run"); return 1; }
    } else {
        o->u = icar_field_f(c, ICAR_F_U); o->v = icar_field_f(c, ICAR_F_V); o->w = icar_field_f(c, ICAR_F_W);
        if (!o->u || !o->v || !o->w) return 1;
    }
    return 0;
}

}  // namespace

// mass_conservative_acceleration (wind.f90:500-511): u = u / u_accel, v = v / v_accel
int icar_mass_conservative_acceleration(icar_hip_ctx *c, int update)
{
    float *u = update ? c->dqdt[ICAR_F_U] : icar_field_f(c, ICAR_F_U), *v = update ? c->dqdt[ICAR_F_V] : icar_field_f(c, ICAR_F_V);
    if (!u || !v) { if (update) icar_set_error("mass_conservative_acceleration(update): upload the u and v dqdt_3d first"); return 1; }
    const float *au = icar_field_f(c, ICAR_F_ZR_U), *av = icar_field_f(c, ICAR_F_ZR_V);
    if (!au || !av) return 1;
    hipLaunchKernelGGL(k_divide, dim3(2048), dim3(256), 0, c->stream, icar_field_count(c, ICAR_F_U), u, au);
    hipLaunchKernelGGL(k_divide, dim3(2048), dim3(256), 0, c->stream, icar_field_count(c, ICAR_F_V), v, av);
    HIPCHK(hipGetLastError());
    icar_winds_changed(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// make_winds_grid_relative (wind.f90:236-287).  The reference works in place with whole-array statements; here the
// rotated mass-grid winds go to a scratch pair and a second pass restaggers from it.  Same operations, same order:
// REAL (a+b)/2, the REAL*DOUBLE products in double rounded once on assignment.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void k_wgr_rotate(Dims d, const float *__restrict__ u, const float *__restrict__ v, const double *__restrict__ st,
                             const double *__restrict__ ct, float *__restrict__ ur, float *__restrict__ vr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, j = blockIdx.z;
    if (i >= d.nx) return;
    const size_t cu = (size_t)i + (size_t)(d.nx + 1) * (k + (size_t)d.nz * j), c = (size_t)d.idx(i, k, j);
    const float uc = (u[cu] + u[cu + 1]) / 2, vc = (v[c] + v[c + d.sj]) / 2;              // :254-255
    const double cs = ct[i + (size_t)d.nx * j], sn = st[i + (size_t)d.nx * j];
    ur[c] = (float)((double)uc * cs - (double)vc * sn);                                    // :260
    vr[c] = (float)((double)vc * cs + (double)uc * sn);                                    // :261
}
__global__ void k_wgr_restagger(Dims d, const float *__restrict__ ur, const float *__restrict__ vr, float *__restrict__ u, float *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, j = blockIdx.z;      // i in 0..nx, j in 0..ny
    const int nx = d.nx, ny = d.ny;
    if (i <= nx && j < ny) {                                                                 // u (nx+1, nz, ny): :270-272
        const float *r = ur + (size_t)nx * (k + (size_t)d.nz * j);
        float val;
        if (i == 0) val = 2 * r[0] - (r[0] + r[1]) / 2;
        else if (i == nx) val = 2 * ((r[nx - 2] + r[nx - 1]) / 2) - ((nx - 2 >= 1) ? (r[nx - 3] + r[nx - 2]) / 2 : 2 * r[0] - (r[0] + r[1]) / 2);
        else val = (r[i - 1] + r[i]) / 2;
        u[(size_t)i + (size_t)(nx + 1) * (k + (size_t)d.nz * j)] = val;
    }
    if (i < nx && j <= ny) {                                                                 // v (nx, nz, ny+1): :274-276
        const size_t sj = (size_t)d.sj; const float *r = vr + (size_t)i + (size_t)nx * k;
        float val;
        if (j == 0) val = 2 * r[0] - (r[0] + r[sj]) / 2;
        else if (j == ny) val = 2 * ((r[(ny - 2) * sj] + r[(ny - 1) * sj]) / 2) - ((ny - 2 >= 1) ? (r[(ny - 3) * sj] + r[(ny - 2) * sj]) / 2 : 2 * r[0] - (r[0] + r[sj]) / 2);
        else val = (r[(j - 1) * sj] + r[j * sj]) / 2;
        v[(size_t)i + (size_t)nx * (k + (size_t)d.nz * j)] = val;
    }
}
}  // namespace

int icar_make_winds_grid_relative(icar_hip_ctx *c, int update)
{
    float *u = update ? c->dqdt[ICAR_F_U] : icar_field_f(c, ICAR_F_U), *v = update ? c->dqdt[ICAR_F_V] : icar_field_f(c, ICAR_F_V);
    if (!u || !v) { if (update) icar_set_error("make_winds_grid_relative(update): upload the u and v dqdt_3d first"); return 1; }
    const double *st = (const double *)icar_field_f(c, ICAR_F_SINTHETA), *ct = (const double *)icar_field_f(c, ICAR_F_COSTHETA);
    if (!st || !ct) return 1;
    if (c->d.nx < 2 || c->d.ny < 2) { icar_set_error("make_winds_grid_relative: tile too small"); return 1; }
    if (!c->wgr_tmp) HIPCHK(hipMalloc(&c->wgr_tmp, 2 * c->n3 * sizeof(float)));
    float *ur = c->wgr_tmp, *vr = c->wgr_tmp + c->n3;
    hipLaunchKernelGGL(k_wgr_rotate, dim3((c->d.nx + 63) / 64, c->d.nz, c->d.ny), dim3(64), 0, c->stream, c->d, u, v, st, ct, ur, vr);
    hipLaunchKernelGGL(k_wgr_restagger, dim3((c->d.nx + 1 + 63) / 64, c->d.nz, c->d.ny + 1), dim3(64), 0, c->stream, c->d, ur, vr, u, v);
    HIPCHK(hipGetLastError());
    if (!update) icar_winds_changed(c);
    return 0;
}

int icar_iterative_winds_correct_w(icar_hip_ctx *c, int update)
{
    Winds q;
    if (pick_winds(c, update, &q)) return 1;
    const float *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    if (!dz) return 1;
    ScopedTimer t(c, "iterative_winds");
    hipLaunchKernelGGL(k_iw_correct_w, dim3((c->d.nx + 63) / 64, c->d.ny), dim3(64), 0, c->stream, c->d, q.w, dz);
    HIPCHK(hipGetLastError());
    if (!update) icar_winds_changed(c);          // domain w rewritten: Courant winds and a prefetched CFL maximum are stale
    return 0;
}

int icar_iterative_winds_sweep(icar_hip_ctx *c, float dx, int nsweeps, int update)
{
    Winds q;
    if (pick_winds(c, update, &q)) return 1;
    const float *ju = icar_field_f(c, ICAR_F_JACOBIAN_U), *jv = icar_field_f(c, ICAR_F_JACOBIAN_V);
    const float *jw = icar_field_f(c, ICAR_F_JACOBIAN_W), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN);
    if (!ju || !jv || !jw || !dz || !jaco) return 1;
    if (!c->iw_adj) HIPCHK(hipMalloc(&c->iw_adj, c->n3 * sizeof(float)));
    ScopedTimer t(c, "iterative_winds");
    const dim3 g((c->d.nx + 255) / 256, c->d.nz, c->d.ny), b(256);
    for (int s = 0; s < nsweeps; ++s) {
        hipLaunchKernelGGL(k_iw_adj, g, b, 0, c->stream, c->d, q.u, q.v, q.w, ju, jv, jw, dz, jaco, dx, c->iw_adj);
        hipLaunchKernelGGL(k_iw_apply, g, b, 0, c->stream, c->d, q.u, q.v, c->iw_adj);
    }
    HIPCHK(hipGetLastError());
    icar_winds_changed(c);
    return 0;
}

int icar_box_copy(icar_hip_ctx *c, int field, int which, int i0, int ni, int j0, int nj, float *buf, bool unpack)
{
    if (field < 0 || field >= ICAR_N_FIELDS || icar_hip_field_elem_size(field) != 4) { icar_set_error("box: REAL(4) fields only"); return 1; }
    const int nx = c->d.nx, nz = c->d.nz, ny = c->d.ny;
    if (icar_field_count(c, field) < (size_t)nx * nz * ny) { icar_set_error("box: 3-D fields only"); return 1; }
    const int X = (field == ICAR_F_U || field == ICAR_F_JACOBIAN_U || field == ICAR_F_DZDX || field == ICAR_F_ZR_U) ? nx + 1 : nx;
    const int Y = (field == ICAR_F_V || field == ICAR_F_JACOBIAN_V || field == ICAR_F_DZDY || field == ICAR_F_ZR_V) ? ny + 1 : ny;
    if (i0 < 0 || ni < 1 || i0 + ni > X || j0 < 0 || nj < 1 || j0 + nj > Y) { icar_set_error("box: range outside the field"); return 1; }
    float *f = which ? c->dqdt[field] : icar_field_f(c, field);
    if (!f) { if (which) icar_set_error("box: dqdt_3d of this field is not on the device"); return 1; }
    ScopedTimer t(c, "halo");
    const dim3 g((ni + 63) / 64, nz, nj), b(64);
    if (unpack) hipLaunchKernelGGL(k_box<true>, g, b, 0, c->stream, X, nz, i0, ni, j0, nj, f, buf);
    else        hipLaunchKernelGGL(k_box<false>, g, b, 0, c->stream, X, nz, i0, ni, j0, nj, f, buf);
    HIPCHK(hipGetLastError());
    return 0;
}
