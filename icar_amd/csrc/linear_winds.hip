// icar_amd/csrc/linear_winds.hip -- rows W2 + W3 of SURVEY.md section 8: linear-theory mountain-wave winds.
//
//   W3  setup_linwinds            src/physics/linear_winds.f90:1180-1225  (buffered terrain, forward FFT, /N, fftshift)
//       add_buffer_topo           :351-418   (host, FP64; sequential in-place smoothing, init only)
//       initialize_linear_theory_data :426-470 (k, l wavenumber axes)
//       linear_perturbation_at_height :181-237, linear_perturbation_constz :239-276
//       initialize_spatial_winds  :596-830   (LUT over dir x spd x N^2 x level, destagger :766-772)
//   W2  spatial_winds             :840-1127  (N^2 field, smoothing, bracket search, 8-corner LUT interpolation)
//
// MI355X design.
//  * FFTs: rocFFT, FP64 complex (like the reference's FFTW plans), batched.  The reference runs one pair of
//    inverse FFTs per (combo, level, sub-layer); the inverse transform is linear, so here the sub-layer sum is
//    taken in spectral space (each term still rounded through the reference's single-precision ifftshift temp,
//    fftshift.f90:221) and ONE pair of FFTs per (combo, level) is done, all levels of a combo in one batched
//    plan.  Equal to the reference order up to FP64 rounding of the FFT itself (1e-16 of the field amplitude),
//    ~5x fewer transforms.
//  * LUT layout in HBM: [nsq][dir][spd][j][k][i] -- one full (i,k,j) field per combo, i fastest.  The reference's
//    (spd,dir,nsq,i,k,j) order makes every lane of a wave gather 8 words from 8 different cache lines; with the
//    field-per-combo order neighbouring lanes (same bracket almost everywhere) read 8 coalesced rows.  At the
//    north-star tile (512x512x40, 720 combos) the two LUTs are 60 GB: resident in the 288 GB of HBM3E, no
//    disk cache needed (the reference's known memory pain point, docs/errors.md:55).
//  * Every image builds all combos for its own tile slice (no coarray scatter :568-590): the whole build is a
//    few seconds of GPU time, cheaper than an all-to-all of the LUT.
//  * W2 is per forcing step, bandwidth/gather bound; running sums of smooth_array are kept sequential (one
//    thread per line) so that the result is bit-identical to the reference's FP64 running sums.
#include "ctx.h"
#include "glibc_flt32.h"
#include <rocfft/rocfft.h>
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstring>

// One batched 2-D FP64 complex transform of the (fftnx, fftny) planes, rocFFT called directly (north star: "the linear-wind
// FFT uses rocFFT"; round 1 went through the hipFFT front end).  A rocFFT plan fixes direction and placement, so a slot
// keeps one plan per use: inverse in place (every LUT entry / perturbation) and forward out of place (the terrain spectrum).
struct RocPlan {
    rocfft_plan inv = nullptr, fwd = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr; size_t work_bytes = 0;
    int batch = 0;
    void destroy()
    {
        if (inv) rocfft_plan_destroy(inv);
        if (fwd) rocfft_plan_destroy(fwd);
        if (info) rocfft_execution_info_destroy(info);
        if (work) hipFree(work);
        inv = fwd = nullptr; info = nullptr; work = nullptr; work_bytes = 0; batch = 0;
    }
};

struct LinWinds {
    icar_hip_lt_options o;
    int nxg = 0, nyg = 0, fftnx = 0, fftny = 0, buffer = 0;   // buffer = lt_options%buffer + 2 (:1205)
    int a0 = 0, b0 = 0;                                        // ims-ids, jms-jds: tile offset in the global domain
    float dx = 0;
    double2 *hhat = nullptr;                                   // domain%terrain_frequency (fftnx, fftny)
    float *k1 = nullptr, *l1 = nullptr;                        // lt_data%k(:,1), lt_data%l(1,:)
    std::vector<float> dirv, spdv, nsqv;
    float *d_vals = nullptr;                                   // dir | spd | nsq values on the device
    RocPlan plan[2];                                           // slot 0: single transforms (either direction); slot 1: the LUT build's batch
    double2 *spec = nullptr; size_t spec_cap = 0;              // [comp][level][fftny][fftnx]
    float *d_z = nullptr; int *d_nsteps = nullptr; int max_steps = 0, nlev = 0;
    float *lut[2] = {nullptr, nullptr}; bool lut_ready = false;
    float *pert[2] = {nullptr, nullptr};                       // hi_u_perturbation (nx+1,nz,ny), hi_v_perturbation (nx,nz,ny+1)
    double *rowmeans = nullptr; float *u1d = nullptr, *v1d = nullptr; int4 *brk = nullptr; float2 *brw = nullptr;
};

static int fftchk(rocfft_status r, const char *what)
{
    if (r == rocfft_status_success) return 0;
    char b[128]; snprintf(b, sizeof b, "rocFFT error %d in %s", (int)r, what);
    icar_set_error(b); return 1;
}
#define FFTCHK(x) do { if (fftchk((x), #x)) return 1; } while (0)

void icar_linwinds_free(icar_hip_ctx *c)
{
    LinWinds *w = c->linwinds;
    if (!w) return;
    for (int i = 0; i < 2; ++i) { w->plan[i].destroy(); if (w->lut[i]) hipFree(w->lut[i]); if (w->pert[i]) hipFree(w->pert[i]); }
    void *p[] = {w->hhat, w->k1, w->l1, w->d_vals, w->spec, w->d_z, w->d_nsteps, w->rowmeans, w->u1d, w->v1d, w->brk, w->brw};
    for (void *q : p) if (q) hipFree(q);
    delete w; c->linwinds = nullptr;
}

// ------------------------------------------------------------------------------------------------ host: add_buffer_topo
// linear_winds.f90:351-418.  buffer_topo is complex(8) with zero imaginary part throughout, so plain doubles
// carry exactly the same values.  terrain(i,j) = t[i + tx*j].
static std::vector<double> add_buffer_topo(const float *t, int tx, int ty, int smooth_window, int b, int &nx, int &ny)
{
    nx = tx + 2 * b; ny = ty + 2 * b;
    float mn = t[0];
    for (size_t n = 0; n < (size_t)tx * ty; ++n) mn = std::min(mn, t[n]);
    std::vector<double> bt((size_t)nx * ny, (double)mn);
    auto B = [&](int i, int j) -> double & { return bt[(size_t)(i - 1) + (size_t)nx * (j - 1)]; };   // 1-based
    auto T = [&](int i, int j) -> float { return t[(size_t)(i - 1) + (size_t)tx * (j - 1)]; };
    for (int j = 1; j <= ty; ++j) for (int i = 1; i <= tx; ++i) B(i + b, j + b) = T(i, j);
    for (int i = 1; i <= b; ++i) {
        const float weight = (float)i / ((float)b * 2);
        const float omw = 1 - weight;
        const int pos = b - i;
        for (int j = 1; j <= ty; ++j) {
            B(pos + 1, j + b) = (double)(T(1, j) * omw + T(tx, j) * weight);
            B(nx - pos, j + b) = (double)(T(1, j) * weight + T(tx, j) * omw);
        }
    }
    for (int i = 1; i <= b; ++i) {
        const float weight = (float)i / ((float)b * 2);
        const double w = (double)weight, omw = (double)(1 - weight);
        const int pos = b - i;
        for (int x = 1; x <= nx; ++x) {
            const double lo = B(x, b + 1), hi = B(x, ny - b);
            B(x, pos + 1) = lo * omw + hi * w;
            B(x, ny - pos) = lo * w + hi * omw;
        }
    }
    if (smooth_window > 0) {
        auto boxmean = [&](int xs, int xe, int ys, int ye) {
            double acc = 0;                               // Fortran SUM of the section, column-major order
            for (int jj = ys; jj <= ye; ++jj) for (int ii = xs; ii <= xe; ++ii) acc += B(ii, jj);
            return acc / (double)((xe - xs + 1) * (ye - ys + 1));
        };
        for (int j = 1; j <= b; ++j) {
            const int window = std::min(j, smooth_window);
            for (int i = 1; i <= nx; ++i) {
                const int xs = std::max(1, i - window), xe = std::min(nx, i + window);
                int ys = std::max(1, b - j + 1 - window), ye = std::min(ny, b - j + 1 + window);
                B(i, b - j + 1) = boxmean(xs, xe, ys, ye);
                ys = std::max(1, ny - (b - j) - window); ye = std::min(ny, ny - (b - j) + window);
                B(i, ny - (b - j)) = boxmean(xs, xe, ys, ye);
            }
            for (int i = 1; i <= ny; ++i) {
                int xs = std::max(1, b - j + 1 - window), xe = std::min(nx, b - j + 1 + window);
                const int ys = std::max(1, i - window), ye = std::min(ny, i + window);
                B(b - j + 1, i) = boxmean(xs, xe, ys, ye);
                xs = std::max(1, nx - (b - j) - window); xe = std::min(nx, nx - (b - j) + window);
                B(nx - (b - j), i) = boxmean(xs, xe, ys, ye);
            }
        }
    }
    return bt;
}

// array_utilities.f90:215-237
static std::vector<float> linear_space(float vmin, float vmax, int n)
{
    std::vector<float> v(n);
    for (int i = 1; i <= n; ++i) v[i - 1] = ((float)i - 1.0f) / (float)((float)n - 1.0f) * (vmax - vmin) + vmin;
    return v;
}

// ------------------------------------------------------------------------------------------------ W3 kernels
__device__ __forceinline__ int shift_src(int i0, int n)       // fftshift.f90: ii = mod(i+(n+1)/2, n), 0 -> n (1-based)
{
    int ii = (i0 + 1 + (n + 1) / 2) % n;
    if (ii == 0) ii = n;
    return ii - 1;
}

// terrain_frequency = fftshift(FFT(terrain) / (nx*ny))  (:1218-1223); tmp(ii,jj) = cmplx(f(i,j)) in single precision
__global__ void k_lt_norm_shift(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= nx) return;
    const double n = (double)(nx * ny);
    const double2 v = in[(size_t)i + (size_t)nx * j];
    const int ii = shift_src(i, nx), jj = shift_src(j, ny);
    out[(size_t)ii + (size_t)nx * jj] = make_double2((double)(float)(v.x / n), (double)(float)(v.y / n));
}

// linear_perturbation_at_height (:198-231) for every sub-layer height of levels [lev0, lev0+nlev), summed over the
// sub-layers of a level (see header), written at the ifftshift-ed position.  spec: u planes [0,nlev), v planes [vplane0, vplane0+nlev).
__global__ void __launch_bounds__(64)
k_lt_spectral(const double2 *__restrict__ hhat, const float *__restrict__ k1, const float *__restrict__ l1, int nx, int ny,
              float U, float V, float Nsq, const float *__restrict__ zs, const int *__restrict__ nsteps, int max_steps,
              int lev0, int nlev, double2 *__restrict__ spec, int vplane0)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
    if (i >= nx) return;
    const int ii = shift_src(i, nx), jj = shift_src(j, ny);          // tmp(i,j) = uhat(ii,jj)
    const float k = k1[ii], l = l1[jj];
    float kl = k * k + l * l;
    if (kl == 0.0f) kl = 1e-15f;                                     // SMALL_VALUE :470
    float sig = U * k + V * l;
    if (sig == 0.0f) sig = 1e-15f;
    const double denom = (double)(sig * sig);                        // sig**2 evaluated in real(4)
    const double msq = ((double)Nsq / denom) * (double)kl;           // >= 0: the evanescent branch :212-214 is unreachable
    double mr = sqrt(msq);
    if (sig < 0) mr = -mr;
    const double q = (double)kl / ((0.0 - mr) * (double)sig);        // kl / ((0-m)*sig)
    const double2 h = hhat[(size_t)ii + (size_t)nx * jj];
    const double ar = -h.y, ai = h.x;                                // imaginary_number * fourier_terrain
    const size_t plane = (size_t)nx * ny, o = (size_t)i + (size_t)nx * j;
    for (int lv = 0; lv < nlev; ++lv) {
        const int ns = nsteps[lev0 + lv];
        double ur = 0, ui = 0, vr = 0, vi = 0;
        for (int s = 0; s < ns; ++s) {
            const double theta = mr * (double)zs[(size_t)(lev0 + lv) * max_steps + s];
            double sn, cs;
            sincos(theta, &sn, &cs);
            const double tr = ar * cs - ai * sn, ti = ar * sn + ai * cs;
            const double ir = tr / q, im = ti / q;
            ur += (double)(float)((double)k * ir); ui += (double)(float)((double)k * im);   // single-precision ifftshift temp
            vr += (double)(float)((double)l * ir); vi += (double)(float)((double)l * im);
        }
        spec[(size_t)lv * plane + o] = make_double2(ur, ui);
        spec[(size_t)(vplane0 + lv) * plane + o] = make_double2(vr, vi);
    }
}

// temporary_u / temporary_v (:766-772) of the tile slice, into the LUT plane of this combo.
__global__ void k_lt_destagger(const double2 *__restrict__ spec, int fnx, int fny, int nlev_chunk, int lev0, const int *__restrict__ nsteps,
                               int buffer, int a0, int b0, int nx, int nz, int ny, float *__restrict__ ulut, float *__restrict__ vlut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, lv = blockIdx.z;
    const size_t plane = (size_t)fnx * fny;
    const double n = (double)nsteps[lev0 + lv];
    const double2 *su = spec + (size_t)lv * plane, *sv = spec + (size_t)(nlev_chunk + lv) * plane;
    if (i <= nx && j < ny) {
        const int fi = buffer + a0 + i - 1, fj = buffer + b0 + j;
        const double s = su[(size_t)fi + (size_t)fnx * fj].x / n + su[(size_t)fi + 1 + (size_t)fnx * fj].x / n;
        ulut[(size_t)i + (size_t)(nx + 1) * ((size_t)(lev0 + lv) + (size_t)nz * j)] = (float)s / 2.0f;
    }
    if (i < nx && j <= ny) {
        const int fi = buffer + a0 + i, fj = buffer + b0 + j - 1;
        const double s = sv[(size_t)fi + (size_t)fnx * fj].x / n + sv[(size_t)fi + (size_t)fnx * (fj + 1)].x / n;
        vlut[(size_t)i + (size_t)nx * ((size_t)(lev0 + lv) + (size_t)nz * j)] = (float)s / 2.0f;
    }
}

// linear_perturbation_varyingz (:316-343) + destagger: the sub-layer solutions of one model level are weighted in
// physical space by the fraction of [current_z +- step/2] that lies inside each column's layer.
struct VaryZ {
    const float *zb, *zt;      // global z_bottom / z_top (nxg, nz, nyg)
    int nxg, nzg, nyg, level;
    float start_z, end_z, step_size;
    const float *cz; int nsub; // sub-layer centre heights of this level
};

__device__ __forceinline__ double lt_varying_point(const double2 *__restrict__ planes, size_t plane, size_t p, int fi, int fj, int buffer, const VaryZ &v)
{
    // internal_z_top / internal_z_bottom (:302-305): the field sits at 1-based (buffer : buffer+n-1)
    const int gi = fi - (buffer - 1), gj = fj - (buffer - 1);
    float izt = v.end_z, izb = v.start_z;
    if (gi >= 0 && gi < v.nxg && gj >= 0 && gj < v.nyg) {
        const size_t g = (size_t)gi + (size_t)v.nxg * ((size_t)v.level + (size_t)v.nzg * gj);
        izt = v.zt[g]; izb = v.zb[g];
    }
    const float half = v.step_size / 2;
    float layer_count = 0;
    double acc = 0;
    for (int s = 0; s < v.nsub; ++s) {
        const float cz = v.cz[s];
        const float frac = fmaxf(0.0f, ((fminf(half, cz - izb) + fminf(0.0f, izt - cz)) + fminf(half, izt - cz)) + fminf(0.0f, cz - izb)) / v.step_size;
        layer_count = layer_count + frac;
        acc = acc + planes[(size_t)s * plane + p].x * (double)frac;
    }
    return acc / (double)layer_count;
}

__global__ void k_lt_destagger_varying(const double2 *__restrict__ spec, int fnx, int fny, int maxsub, VaryZ vz,
                                       int buffer, int a0, int b0, int nx, int nz, int ny, float *__restrict__ ulut, float *__restrict__ vlut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    const size_t plane = (size_t)fnx * fny;
    const double2 *su = spec, *sv = spec + (size_t)maxsub * plane;
    if (i <= nx && j < ny) {
        const int fi = buffer + a0 + i - 1, fj = buffer + b0 + j;
        const double s = lt_varying_point(su, plane, (size_t)fi + (size_t)fnx * fj, fi, fj, buffer, vz)
                       + lt_varying_point(su, plane, (size_t)fi + 1 + (size_t)fnx * fj, fi + 1, fj, buffer, vz);
        ulut[(size_t)i + (size_t)(nx + 1) * ((size_t)vz.level + (size_t)nz * j)] = (float)s / 2.0f;
    }
    if (i < nx && j <= ny) {
        const int fi = buffer + a0 + i, fj = buffer + b0 + j - 1;
        const double s = lt_varying_point(sv, plane, (size_t)fi + (size_t)fnx * fj, fi, fj, buffer, vz)
                       + lt_varying_point(sv, plane, (size_t)fi + (size_t)fnx * (fj + 1), fi, fj + 1, buffer, vz);
        vlut[(size_t)i + (size_t)nx * ((size_t)vz.level + (size_t)nz * j)] = (float)s / 2.0f;
    }
}

__global__ void k_lt_real_parts(const double2 *__restrict__ spec, size_t plane, int nsteps, double *__restrict__ u, double *__restrict__ v)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= plane) return;
    u[t] = spec[t].x / (double)nsteps;
    v[t] = spec[plane + t].x / (double)nsteps;
}

// LUT layout conversion: reference (s,d,n,cell) <-> device (combo, cell), one j-row slab at a time
__global__ void k_lut_transpose(float *__restrict__ dev, float *__restrict__ ref, int ncombo, size_t slab_cells, size_t field_cells, size_t cell0, int to_dev)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= slab_cells * ncombo) return;
    if (to_dev) {       // coalesced writes to dev
        const size_t cell = t % slab_cells; const int c = (int)(t / slab_cells);
        dev[(size_t)c * field_cells + cell0 + cell] = ref[(size_t)c + (size_t)ncombo * cell];
    } else {            // coalesced writes to ref
        const int c = (int)(t % ncombo); const size_t cell = t / ncombo;
        ref[t] = dev[(size_t)c * field_cells + cell0 + cell];
    }
}

// ------------------------------------------------------------------------------------------------ W3 host
static int ensure_plan(LinWinds *w, int slot, int batch, hipStream_t s)
{
    static bool setup_done = false;
    if (!setup_done) { FFTCHK(rocfft_setup()); setup_done = true; }
    RocPlan &P = w->plan[slot];
    if (P.batch != batch) {
        P.destroy();
        const size_t len[2] = {(size_t)w->fftnx, (size_t)w->fftny};          // fastest dimension first
        FFTCHK(rocfft_plan_create(&P.inv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 2, len, (size_t)batch, nullptr));
        FFTCHK(rocfft_plan_create(&P.fwd, rocfft_placement_notinplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 2, len, (size_t)batch, nullptr));
        FFTCHK(rocfft_execution_info_create(&P.info));
        size_t wi = 0, wf = 0;
        FFTCHK(rocfft_plan_get_work_buffer_size(P.inv, &wi));
        FFTCHK(rocfft_plan_get_work_buffer_size(P.fwd, &wf));
        P.work_bytes = wi > wf ? wi : wf;
        if (P.work_bytes) {
            HIPCHK(hipMalloc(&P.work, P.work_bytes));
            FFTCHK(rocfft_execution_info_set_work_buffer(P.info, P.work, P.work_bytes));
        }
        P.batch = batch;
    }
    FFTCHK(rocfft_execution_info_set_stream(P.info, s));
    return 0;
}
static int fft_inverse_inplace(LinWinds *w, int slot, double2 *buf)
{
    void *io[1] = {buf};
    return fftchk(rocfft_execute(w->plan[slot].inv, io, nullptr, w->plan[slot].info), "rocfft_execute(inverse)");
}
static int fft_forward(LinWinds *w, int slot, double2 *in, double2 *out)
{
    void *i_[1] = {in}, *o_[1] = {out};
    return fftchk(rocfft_execute(w->plan[slot].fwd, i_, o_, w->plan[slot].info), "rocfft_execute(forward)");
}

static int ensure_spec(LinWinds *w, size_t elems)
{
    if (w->spec_cap >= elems) return 0;
    if (w->spec) hipFree(w->spec);
    w->spec = nullptr; w->spec_cap = 0;
    HIPCHK(hipMalloc(&w->spec, elems * sizeof(double2)));
    w->spec_cap = elems;
    return 0;
}

int icar_linwinds_setup_run(icar_hip_ctx *c, const icar_hip_lt_options *o, const float *terrain, int nxg, int nyg, int ids, int jds, float dx)
{
    if (o->buffer < 1) { icar_set_error("linwinds_setup: lt_options%buffer must be >= 1"); return 1; }
    if (o->n_dir_values < 2 || o->n_spd_values < 2 || o->n_nsq_values < 2) { icar_set_error("linwinds_setup: LUT axes need >= 2 values"); return 1; }
    const int a0 = c->ims - ids, b0 = c->jms - jds;
    if (a0 < 0 || b0 < 0 || a0 + c->d.nx > nxg || b0 + c->d.ny > nyg) { icar_set_error("linwinds_setup: tile lies outside the global terrain"); return 1; }
    icar_linwinds_free(c);
    LinWinds *w = new LinWinds();
    c->linwinds = w;
    w->o = *o; w->nxg = nxg; w->nyg = nyg; w->dx = dx; w->a0 = a0; w->b0 = b0;
    int nx1, ny1, nx2, ny2;
    std::vector<double> first = add_buffer_topo(terrain, nxg, nyg, 5, o->buffer, nx1, ny1);            // :1201
    std::vector<float> firstf(first.size());
    for (size_t n = 0; n < first.size(); ++n) firstf[n] = (float)first[n];                             // real(real(...)) :1203
    std::vector<double> second = add_buffer_topo(firstf.data(), nx1, ny1, 0, 2, nx2, ny2);
    w->buffer = 2 + o->buffer;                                                                         // :1205
    w->fftnx = nx2; w->fftny = ny2;
    const size_t plane = (size_t)nx2 * ny2;
    std::vector<double2> cplx(plane);
    for (size_t n = 0; n < plane; ++n) cplx[n] = make_double2(second[n], 0.0);
    HIPCHK(hipMalloc(&w->hhat, plane * sizeof(double2)));
    if (ensure_spec(w, plane * 2)) return 1;
    HIPCHK(hipMemcpyAsync(w->spec, cplx.data(), plane * sizeof(double2), hipMemcpyHostToDevice, c->stream));
    if (ensure_plan(w, 0, 1, c->stream)) return 1;
    if (fft_forward(w, 0, w->spec, w->spec + plane)) return 1;   // :1216-1218
    k_lt_norm_shift<<<dim3((nx2 + 63) / 64, ny2), 64, 0, c->stream>>>(w->spec + plane, w->hhat, nx2, ny2);
    HIPCHK(hipGetLastError());
    // wavenumber axes :447-462
    const float pi = 3.1415927f;
    std::vector<float> k1(nx2), l1(ny2);
    const float offset = pi / dx;
    float gain = 2 * offset / (float)(nx2 - 1);
    for (int i = 1; i <= nx2; ++i) k1[i - 1] = (float)(i - 1) * gain - offset;
    gain = 2 * offset / (float)(ny2 - 1);
    for (int i = 1; i <= ny2; ++i) l1[i - 1] = (float)(i - 1) * gain - offset;
    HIPCHK(hipMalloc(&w->k1, nx2 * sizeof(float))); HIPCHK(hipMalloc(&w->l1, ny2 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(w->k1, k1.data(), nx2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(w->l1, l1.data(), ny2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // LUT axes :648-650
    w->dirv = linear_space(o->dirmin, o->dirmax, o->n_dir_values);
    w->nsqv = linear_space(o->nsqmin, o->nsqmax, o->n_nsq_values);
    w->spdv = linear_space(o->spdmin, o->spdmax, o->n_spd_values);
    std::vector<float> vals; vals.insert(vals.end(), w->dirv.begin(), w->dirv.end());
    vals.insert(vals.end(), w->spdv.begin(), w->spdv.end()); vals.insert(vals.end(), w->nsqv.begin(), w->nsqv.end());
    HIPCHK(hipMalloc(&w->d_vals, vals.size() * sizeof(float)));
    HIPCHK(hipMemcpyAsync(w->d_vals, vals.data(), vals.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // hi_u_perturbation / hi_v_perturbation = 0 :1263-1268
    for (int comp = 0; comp < 2; ++comp) {
        const size_t cnt = icar_field_count(c, comp == 0 ? ICAR_F_U : ICAR_F_V);
        HIPCHK(hipMalloc(&w->pert[comp], cnt * sizeof(float)));
        HIPCHK(hipMemsetAsync(w->pert[comp], 0, cnt * sizeof(float), c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int upload_levels(icar_hip_ctx *c, LinWinds *w, const float *zb, const float *zt, int nlev, float minimum_step)
{
    // linear_perturbation_constz :255-262: n_steps, step_size, current_z sequence (all real(4))
    std::vector<int> ns(nlev);
    int mx = 1;
    for (int z = 0; z < nlev; ++z) {
        ns[z] = std::max(1, (int)std::ceil((zt[z] - zb[z]) / minimum_step));
        mx = std::max(mx, ns[z]);
    }
    std::vector<float> zs((size_t)nlev * mx, 0.0f);
    for (int z = 0; z < nlev; ++z) {
        const float step_size = (zt[z] - zb[z]) / (float)ns[z];
        float current_z = zb[z] + step_size / 2;
        for (int s = 0; s < ns[z]; ++s) { zs[(size_t)z * mx + s] = current_z; current_z = current_z + step_size; }
    }
    if (w->d_z) hipFree(w->d_z);
    if (w->d_nsteps) hipFree(w->d_nsteps);
    w->d_z = nullptr; w->d_nsteps = nullptr;
    HIPCHK(hipMalloc(&w->d_z, zs.size() * sizeof(float))); HIPCHK(hipMalloc(&w->d_nsteps, nlev * sizeof(int)));
    HIPCHK(hipMemcpy(w->d_z, zs.data(), zs.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->d_nsteps, ns.data(), nlev * sizeof(int), hipMemcpyHostToDevice));
    w->max_steps = mx; w->nlev = nlev;
    return 0;
}

int icar_linear_perturbation_run(icar_hip_ctx *c, float U, float V, float Nsq, float zb, float zt, float minimum_step, double *u_out, double *v_out)
{
    LinWinds *w = c->linwinds;
    if (!w) { icar_set_error("linear_perturbation: call icar_hip_linwinds_setup first"); return 1; }
    const size_t plane = (size_t)w->fftnx * w->fftny;
    if (U == 0 && V == 0) {                                             // :248-252
        memset(u_out, 0, plane * sizeof(double)); memset(v_out, 0, plane * sizeof(double));
        return 0;
    }
    if (upload_levels(c, w, &zb, &zt, 1, minimum_step)) return 1;
    if (ensure_spec(w, plane * 4)) return 1;
    if (ensure_plan(w, 0, 2, c->stream)) return 1;
    k_lt_spectral<<<dim3((w->fftnx + 63) / 64, w->fftny), 64, 0, c->stream>>>(w->hhat, w->k1, w->l1, w->fftnx, w->fftny, U, V, Nsq,
                                                                                   w->d_z, w->d_nsteps, w->max_steps, 0, 1, w->spec, 1);
    HIPCHK(hipGetLastError());
    if (fft_inverse_inplace(w, 0, w->spec)) return 1;
    double *du = (double *)(w->spec + 2 * plane), *dv = du + plane;
    int ns1 = 0;
    HIPCHK(hipMemcpy(&ns1, w->d_nsteps, sizeof(int), hipMemcpyDeviceToHost));
    k_lt_real_parts<<<(unsigned)((plane + 255) / 256), 256, 0, c->stream>>>(w->spec, plane, ns1, du, dv);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(u_out, du, plane * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(v_out, dv, plane * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int alloc_luts(icar_hip_ctx *c, LinWinds *w)
{
    const size_t ncombo = (size_t)w->o.n_dir_values * w->o.n_spd_values * w->o.n_nsq_values;
    for (int comp = 0; comp < 2; ++comp) {
        if (w->lut[comp]) continue;
        const size_t bytes = ncombo * icar_field_count(c, comp == 0 ? ICAR_F_U : ICAR_F_V) * sizeof(float);
        if (hipMalloc(&w->lut[comp], bytes) != hipSuccess) {
            char b[160]; snprintf(b, sizeof b, "linwinds: cannot allocate %.1f GB for the %c LUT", bytes / 1e9, comp ? 'v' : 'u');
            icar_set_error(b); return 1;
        }
        HIPCHK(hipMemsetAsync(w->lut[comp], 0, bytes, c->stream));
    }
    return 0;
}

// initialize_spatial_winds :596-830, constant-z branch (:750-762), all LUT entries for this image's tile
int icar_linwinds_build_lut_run(icar_hip_ctx *c, const float *zb, const float *zt, int nlev)
{
    LinWinds *w = c->linwinds;
    if (!w) { icar_set_error("linwinds_build_lut: call icar_hip_linwinds_setup first"); return 1; }
    if (nlev != c->d.nz) { icar_set_error("linwinds_build_lut: need one layer per model level"); return 1; }
    ScopedTimer t(c, "lt_lut");
    if (upload_levels(c, w, zb, zt, nlev, w->o.minimum_layer_size)) return 1;
    if (alloc_luts(c, w)) return 1;
    const size_t plane = (size_t)w->fftnx * w->fftny;
    int chunk = nlev;                                                   // levels per batched FFT, spec <= 4 GB
    while (chunk > 1 && (size_t)chunk * 2 * plane * sizeof(double2) > ((size_t)4 << 30)) chunk = (chunk + 1) / 2;
    if (ensure_spec(w, (size_t)chunk * 2 * plane)) return 1;
    const int nd = w->o.n_dir_values, ns = w->o.n_spd_values, nn = w->o.n_nsq_values;
    const int nx = c->d.nx, nz = c->d.nz, ny = c->d.ny;
    const size_t ucells = icar_field_count(c, ICAR_F_U), vcells = icar_field_count(c, ICAR_F_V);
    for (int ijk = 0; ijk < nd * ns * nn; ++ijk) {                      // :704-709
        const int ik = ijk / nn, j = ijk % nn, i = ik / ns, k = ik % ns;
        const float u = sinf(w->dirv[i]) * w->spdv[k];                  // calc_u / calc_v, atm_utilities.f90:373-391
        const float v = cosf(w->dirv[i]) * w->spdv[k];
        const float nsq = expf(w->nsqv[j]);
        if (u == 0 && v == 0) continue;                                 // perturbation is zero (:248), LUT planes stay 0
        const size_t combo = (size_t)k + (size_t)ns * ((size_t)i + (size_t)nd * j);
        for (int lev0 = 0; lev0 < nlev; lev0 += chunk) {
            const int nl = std::min(chunk, nlev - lev0);
            const int slot = (nl == chunk) ? 0 : 1;
            if (ensure_plan(w, slot, 2 * nl, c->stream)) return 1;
            k_lt_spectral<<<dim3((w->fftnx + 63) / 64, w->fftny), 64, 0, c->stream>>>(w->hhat, w->k1, w->l1, w->fftnx, w->fftny, u, v, nsq,
                                                                                           w->d_z, w->d_nsteps, w->max_steps, lev0, nl, w->spec, nl);
            if (fft_inverse_inplace(w, slot, w->spec)) return 1;
            k_lt_destagger<<<dim3((nx + 1 + 63) / 64, ny + 1, nl), 64, 0, c->stream>>>(w->spec, w->fftnx, w->fftny, nl, lev0, w->d_nsteps, w->buffer,
                                                                                        w->a0, w->b0, nx, nz, ny,
                                                                                        w->lut[0] + combo * ucells, w->lut[1] + combo * vcells);
        }
    }
    HIPCHK(hipGetLastError());
    w->lut_ready = true;
    return 0;
}

// initialize_spatial_winds, space_varying_dz branch (:738-748): zb3/zt3 = global_z_interface - global_terrain
// (+ global_dz_interface), (nx_global, nz, ny_global) Fortran order.
int icar_linwinds_build_lut_varying_run(icar_hip_ctx *c, const float *zb3, const float *zt3, int nlev)
{
    LinWinds *w = c->linwinds;
    if (!w) { icar_set_error("linwinds_build_lut_varying: call icar_hip_linwinds_setup first"); return 1; }
    if (nlev != c->d.nz) { icar_set_error("linwinds_build_lut_varying: need one layer per model level"); return 1; }
    ScopedTimer t(c, "lt_lut");
    const int nxg = w->nxg, nyg = w->nyg;
    // per level: start_z, end_z, step_size and the current_z sequence (:297-314), REAL(4)
    std::vector<float> start_z(nlev), end_z(nlev), step(nlev);
    std::vector<std::vector<float>> cz(nlev);
    size_t maxsub = 1;
    for (int z = 0; z < nlev; ++z) {
        float mn = zb3[(size_t)nxg * z], mx = zt3[(size_t)nxg * z], md = mx - mn;
        for (int j = 0; j < nyg; ++j) for (int i = 0; i < nxg; ++i) {
            const size_t g = (size_t)i + (size_t)nxg * ((size_t)z + (size_t)nlev * j);
            mn = std::min(mn, zb3[g]); mx = std::max(mx, zt3[g]); md = std::min(md, zt3[g] - zb3[g]);
        }
        start_z[z] = mn; end_z[z] = mx; step[z] = std::min(w->o.minimum_layer_size, md);
        if (!(step[z] > 0)) { icar_set_error("linwinds_build_lut_varying: a layer has zero or negative thickness"); return 1; }
        float current_z = mn + step[z] / 2;
        while (current_z < mx) { cz[z].push_back(current_z); current_z = current_z + step[z]; }
        if (cz[z].size() > 4096) { icar_set_error("linwinds_build_lut_varying: more than 4096 sub-layers in one level"); return 1; }
        maxsub = std::max(maxsub, cz[z].size());
    }
    std::vector<float> flat; std::vector<int> off(nlev);
    for (int z = 0; z < nlev; ++z) { off[z] = (int)flat.size(); flat.insert(flat.end(), cz[z].begin(), cz[z].end()); }
    std::vector<int> ones(flat.size(), 1);
    if (w->d_z) hipFree(w->d_z);
    if (w->d_nsteps) hipFree(w->d_nsteps);
    w->d_z = nullptr; w->d_nsteps = nullptr;
    HIPCHK(hipMalloc(&w->d_z, std::max<size_t>(1, flat.size()) * sizeof(float))); HIPCHK(hipMalloc(&w->d_nsteps, std::max<size_t>(1, flat.size()) * sizeof(int)));
    HIPCHK(hipMemcpy(w->d_z, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->d_nsteps, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice));
    w->max_steps = 1; w->nlev = (int)flat.size();
    float *dzb = nullptr, *dzt = nullptr;
    const size_t n3g = (size_t)nxg * nlev * nyg;
    HIPCHK(hipMalloc(&dzb, n3g * sizeof(float))); HIPCHK(hipMalloc(&dzt, n3g * sizeof(float)));
    HIPCHK(hipMemcpy(dzb, zb3, n3g * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dzt, zt3, n3g * sizeof(float), hipMemcpyHostToDevice));
    int rc = 0;
    do {
        if ((rc = alloc_luts(c, w))) break;
        const size_t plane = (size_t)w->fftnx * w->fftny;
        if ((rc = ensure_spec(w, maxsub * 2 * plane))) break;
        if ((rc = ensure_plan(w, 0, (int)(2 * maxsub), c->stream))) break;
        HIPCHK(hipMemsetAsync(w->spec, 0, maxsub * 2 * plane * sizeof(double2), c->stream));
        const int nd = w->o.n_dir_values, ns = w->o.n_spd_values, nn = w->o.n_nsq_values;
        const int nx = c->d.nx, nz = c->d.nz, ny = c->d.ny;
        const size_t ucells = icar_field_count(c, ICAR_F_U), vcells = icar_field_count(c, ICAR_F_V);
        for (int ijk = 0; ijk < nd * ns * nn && !rc; ++ijk) {
            const int ik = ijk / nn, j = ijk % nn, i = ik / ns, k = ik % ns;
            const float u = sinf(w->dirv[i]) * w->spdv[k], v = cosf(w->dirv[i]) * w->spdv[k], nsq = expf(w->nsqv[j]);
            if (u == 0 && v == 0) continue;
            const size_t combo = (size_t)k + (size_t)ns * ((size_t)i + (size_t)nd * j);
            for (int z = 0; z < nlev; ++z) {
                const int nsub = (int)cz[z].size();
                if (nsub == 0) continue;
                // u planes at [0, nsub), v planes at [maxsub, maxsub+nsub): call the spectral kernel once per component block
                k_lt_spectral<<<dim3((w->fftnx + 63) / 64, w->fftny), 64, 0, c->stream>>>(w->hhat, w->k1, w->l1, w->fftnx, w->fftny, u, v, nsq,
                                                                                               w->d_z, w->d_nsteps, 1, off[z], nsub, w->spec, (int)maxsub);
                if (fft_inverse_inplace(w, 0, w->spec)) { rc = 1; break; }
                VaryZ vz{dzb, dzt, nxg, nlev, nyg, z, start_z[z], end_z[z], step[z], w->d_z + off[z], nsub};
                k_lt_destagger_varying<<<dim3((nx + 1 + 63) / 64, ny + 1), 64, 0, c->stream>>>(w->spec, w->fftnx, w->fftny, (int)maxsub, vz, w->buffer,
                                                                                                w->a0, w->b0, nx, nz, ny,
                                                                                                w->lut[0] + combo * ucells, w->lut[1] + combo * vcells);
            }
        }
        if (!rc && icar_hip_check(hipGetLastError(), "linwinds_build_lut_varying kernels")) rc = 1;
        if (!rc && icar_hip_check(hipStreamSynchronize(c->stream), "linwinds_build_lut_varying sync")) rc = 1;
    } while (0);
    hipFree(dzb); hipFree(dzt);
    if (!rc) w->lut_ready = true;
    return rc;
}

int icar_linwinds_lut_copy(icar_hip_ctx *c, int comp, float *host, int to_dev)
{
    LinWinds *w = c->linwinds;
    if (!w) { icar_set_error("linwinds LUT: call icar_hip_linwinds_setup first"); return 1; }
    if (comp < 0 || comp > 1) { icar_set_error("linwinds LUT: component must be 0 (u) or 1 (v)"); return 1; }
    if (to_dev) { if (alloc_luts(c, w)) return 1; }
    else if (!w->lut[comp]) { icar_set_error("linwinds LUT: not built"); return 1; }
    const int ncombo = w->o.n_dir_values * w->o.n_spd_values * w->o.n_nsq_values;
    const size_t cells = icar_field_count(c, comp == 0 ? ICAR_F_U : ICAR_F_V);
    const int nrows = comp == 0 ? c->d.ny : c->d.ny + 1;
    const size_t slab = cells / nrows;                                   // one j-row: (nx[+1]) * nz cells
    float *stage = nullptr;
    HIPCHK(hipMalloc(&stage, slab * ncombo * sizeof(float)));
    for (int r = 0; r < nrows; ++r) {
        float *h = host + (size_t)r * slab * ncombo;
        const size_t tot = slab * ncombo;
        if (to_dev) HIPCHK(hipMemcpyAsync(stage, h, tot * sizeof(float), hipMemcpyHostToDevice, c->stream));
        k_lut_transpose<<<(unsigned)((tot + 255) / 256), 256, 0, c->stream>>>(w->lut[comp], stage, ncombo, slab, cells, (size_t)r * slab, to_dev);
        if (!to_dev) HIPCHK(hipMemcpyAsync(h, stage, tot * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    hipFree(stage);
    if (to_dev) w->lut_ready = true;
    return 0;
}

// one LUT entry (spd k, dir i, nsq j: 0-based) of component comp as the whole (nx[+1], nz, ny[+1]) field it is on the device
int icar_linwinds_lut_entry(icar_hip_ctx *c, int comp, int k, int i, int j, float *host)
{
    LinWinds *w = c->linwinds;
    if (!w || comp < 0 || comp > 1 || !w->lut[comp]) { icar_set_error("linwinds LUT entry: LUT not built / bad component"); return 1; }
    const int nd = w->o.n_dir_values, ns = w->o.n_spd_values, nn = w->o.n_nsq_values;
    if (k < 0 || k >= ns || i < 0 || i >= nd || j < 0 || j >= nn) { icar_set_error("linwinds LUT entry: index outside the LUT axes"); return 1; }
    const size_t cells = icar_field_count(c, comp == 0 ? ICAR_F_U : ICAR_F_V);
    const size_t combo = (size_t)k + (size_t)ns * ((size_t)i + (size_t)nd * j);
    HIPCHK(hipMemcpyAsync(host, w->lut[comp] + combo * cells, cells * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_linwinds_pert_copy(icar_hip_ctx *c, int comp, float *host, int to_dev)
{
    LinWinds *w = c->linwinds;
    if (!w || comp < 0 || comp > 1) { icar_set_error("linwinds perturbation: not set up / bad component"); return 1; }
    const size_t bytes = icar_field_count(c, comp == 0 ? ICAR_F_U : ICAR_F_V) * sizeof(float);
    if (to_dev) HIPCHK(hipMemcpyAsync(w->pert[comp], host, bytes, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(host, w->pert[comp], bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_linwinds_terrain_frequency(icar_hip_ctx *c, double *out, size_t cap, int *fnx, int *fny)
{
    LinWinds *w = c->linwinds;
    if (!w) { icar_set_error("linwinds: not set up"); return 1; }
    if (fnx) *fnx = w->fftnx;
    if (fny) *fny = w->fftny;
    if (!out) return 0;
    const size_t plane = (size_t)w->fftnx * w->fftny;
    if (cap < plane) { icar_set_error("linwinds_terrain_frequency: buffer too small"); return 1; }
    HIPCHK(hipMemcpy(out, w->hhat, plane * sizeof(double2), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------ W2 kernels
struct LtDev {
    int variable_N, smooth_nsq;
    float N_squared, max_stability, min_stability, linear_contribution, linear_update_fraction;
    int nd, ns, nn;
    const float *dirv, *spdv, *nsqv;
};

// REAL(4) log / exp / atan as the compiled reference evaluates them: the C library's logf / expf / atanf (glibc_flt32.h)
__device__ __forceinline__ float w_logf(float x) { return gf_logf(x); }
__device__ __forceinline__ float w_expf(float x) { return gf_expf(x); }
__device__ __forceinline__ float w_atanf(float x) { return gf_atanf(x); }

#define LW_PI 3.1415927f
#define LW_LH 2260000.0f
#define LW_RD 287.058f
#define LW_RW 461.5f
#define LW_CP 1012.0f
#define LW_G 9.81f

__device__ float lw_sat_lapse(float T, float mr)                  // atm_utilities.f90:401-410
{
    const float L = LW_LH;
    return LW_G * ((1 + (L * mr) / (LW_RD * T)) / (LW_CP + (L * L * mr * (LW_RD / LW_RW)) / (LW_RD * T * T)));
}

__device__ float lw_stability(const LtDev &o, float th_top, float th_bot, float pii_top, float pii_bot, float z_top, float z_bot,
                              float qv_top, float qv_bot, float qc)   // atm_utilities.f90:417-467
{
    if (qc < 1e-7f) {
        if (o.variable_N) return LW_G * (w_logf(th_top) - w_logf(th_bot)) / (z_top - z_bot);
        return o.N_squared;
    }
    if (!o.variable_N) return o.N_squared / 10.0f;
    const float t_top = th_top * pii_top, t_bot = th_bot * pii_bot;
    const float t = (t_top + t_bot) / 2, qv = (qv_top + qv_bot) / 2, dz = z_top - z_bot;
    const float sat_lapse = lw_sat_lapse(t, qv);
    return (LW_G / t) * ((t_top - t_bot) / dz + sat_lapse) * (1 + (LW_LH * qv) / (LW_RD * t))
           - (LW_G / (1 + qv + qc) * (qv_top - qv_bot) / dz);
}

// N^2 per cell + log (:906-957)
__global__ void k_lw_nsq(LtDev o, Dims d, int vsmooth, const float *__restrict__ th, const float *__restrict__ exner, const float *__restrict__ z,
                         const float *__restrict__ qv, const float *__restrict__ qc, const float *__restrict__ qi, const float *__restrict__ qr,
                         const float *__restrict__ qs, float *__restrict__ nsq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y + 1, k = blockIdx.z;
    if (i >= d.nx || j > d.nz) return;
    float val;
    if (o.variable_N) {
        const int top = min(j + vsmooth, d.nz);
        const int bottom = max(1, j - (vsmooth - (top - j)));
        const int c = d.idx(i, j - 1, k);
        float hyd = 0;
        if (qc) hyd = hyd + qc[c];
        if (qi) hyd = hyd + qi[c];
        if (qr) hyd = hyd + qr[c];
        if (qs) hyd = hyd + qs[c];
        const int cb = d.idx(i, bottom - 1, k), ct = d.idx(i, top - 1, k);
        val = lw_stability(o, th[cb], th[ct], exner[cb], exner[ct], z[cb], z[ct], qv[cb], qv[ct], hyd);   // argument order of :933-939
        val = fmaxf(o.min_stability, fminf(o.max_stability, val));
    } else {
        val = o.N_squared;
    }
    nsq[d.idx(i, j - 1, k)] = w_logf(val);
}

// vertical box smoothing, in place and bottom-up like :959-973 (levels below j are already smoothed when j is)
__global__ void k_lw_vsmooth(Dims d, int vsmooth, float *__restrict__ nsq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i >= d.nx) return;
    for (int j = 1; j <= d.nz; ++j) {
        const int top = min(j + vsmooth, d.nz);
        const int bottom = max(1, j - (vsmooth - (top - j)));
        float acc = nsq[d.idx(i, j - 1, k)];
        for (int s = bottom; s <= j - 1; ++s) acc = acc + nsq[d.idx(i, s - 1, k)];
        for (int s = j + 1; s <= top; ++s) acc = acc + nsq[d.idx(i, s - 1, k)];
        nsq[d.idx(i, j - 1, k)] = acc / (float)(top - bottom + 1);
    }
}

// smooth_array_3d ydim=3 (array_utilities.f90:355-411), pass 1: running row sums along y, one thread per (i, level)
__global__ void k_lw_rowmeans(Dims d, int w, const float *__restrict__ in, double *__restrict__ rowmeans)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= d.nx) return;
    const int nrow = d.ny, nrows = w * 2 + 1;
    double rs = (double)(in[d.idx(i, j, 0)] * (float)(w + 2));
    const int lim = min(w, nrow);
    for (int r = 2; r <= lim; ++r) rs = rs + (double)in[d.idx(i, j, r - 1)];
    if (w > nrow) rs = rs + (double)(in[d.idx(i, j, nrow - 1)] * (float)(w - nrow));
    for (int k = 1; k <= nrow; ++k) {
        const int starty = max(2, k - w), endy = min(nrow, k + w);
        rs = rs - (double)in[d.idx(i, j, starty - 2)] + (double)in[d.idx(i, j, endy - 1)];
        rowmeans[d.idx(i, j, k - 1)] = rs / (double)nrows;
    }
}

// pass 2: running sum along x, one thread per (level, row)
__global__ void k_lw_colsmooth(Dims d, int w, const double *__restrict__ rowmeans, float *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (j >= d.nz) return;
    const double *rm = rowmeans + d.idx(0, j, k);
    const int ncols = w * 2 + 1;
    double cursum = 0;
    for (int i = 2; i <= w; ++i) cursum += rm[i - 1];
    cursum = cursum + rm[0] * (double)(w + 2);
    for (int i = 1; i <= d.nx; ++i) {
        const int startx = max(2, i - w), endx = min(d.nx, i + w);
        cursum = cursum - rm[startx - 2] + rm[endx - 1];
        out[d.idx(i - 1, j, k)] = (float)(cursum / (double)ncols);
    }
}

__device__ __forceinline__ float lw_weight(const float *dv, int n, int bestpos, int &nextpos, float match)   // calc_weight
{
    if (match < dv[0]) { nextpos = 1; return 1.0f; }
    if (bestpos == n) { nextpos = n; return 1.0f; }
    nextpos = bestpos + 1;
    return (dv[nextpos - 1] - match) / (dv[nextpos - 1] - dv[bestpos - 1]);
}

// column-mean winds + direction / speed brackets (:996-1001, :1040-1060, :1076-1077) for rows [k0, k1] (1-based)
__global__ void k_lw_means(LtDev o, Dims d, int k0, const float *__restrict__ u3d, const float *__restrict__ v3d,
                           float *__restrict__ u1d, float *__restrict__ v1d, int4 *__restrict__ brk, float2 *__restrict__ brw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, k = k0 + blockIdx.y;
    const int nxu = d.nx + 1;
    if (i > nxu) return;
    const int uk = min(k, d.ny), vi = min(i, d.nx);
    float su = 0, sv = 0;
    for (int j = 0; j < d.nz; ++j) su = su + u3d[(size_t)(i - 1) + (size_t)nxu * ((size_t)j + (size_t)d.nz * (uk - 1))];
    for (int j = 0; j < d.nz; ++j) sv = sv + v3d[d.idx(vi - 1, j, k - 1)];
    const float u = su / (float)d.nz, v = sv / (float)d.nz;
    float curdir;                                                    // calc_direction, atm_utilities.f90:334-355
    if (v < 0) curdir = w_atanf(u / v) + LW_PI;
    else if (v == 0) curdir = (u > 0) ? LW_PI / 2.0f : LW_PI * 1.5f;
    else if (u >= 0) curdir = w_atanf(u / v);
    else curdir = w_atanf(u / v) + (2 * LW_PI);
    int dpos = 1, spos = 1, nextd, nexts;
    for (int s = 1; s <= o.nd; ++s) if (curdir > o.dirv[s - 1]) dpos = s;
    const float curspd = sqrtf(u * u + v * v);
    for (int s = 1; s <= o.ns; ++s) if (curspd > o.spdv[s - 1]) spos = s;
    const float dweight = lw_weight(o.dirv, o.nd, dpos, nextd, curdir);
    const float sweight = lw_weight(o.spdv, o.ns, spos, nexts, curspd);
    const size_t t = (size_t)(i - 1) + (size_t)nxu * (k - 1);
    u1d[t] = u; v1d[t] = v;
    brk[t] = make_int4(dpos - 1, nextd - 1, spos - 1, nexts - 1);
    brw[t] = make_float2(dweight, sweight);
}

// 8-corner LUT interpolation, relaxation and update of u3d / v3d (:1062-1115) for rows [k0, ...]
__global__ void k_lw_interp(LtDev o, Dims d, int k0, int winsz, const float *__restrict__ nsq, const int4 *__restrict__ brk,
                            const float2 *__restrict__ brw, const float *__restrict__ ulut, const float *__restrict__ vlut,
                            float *__restrict__ upert, float *__restrict__ vpert, float *__restrict__ u3d, float *__restrict__ v3d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1, k = k0 + blockIdx.z;
    const int nxu = d.nx + 1;
    if (i > nxu || j > d.nz) return;
    const int uk = min(k, d.ny), vi = min(i, d.nx);
    const int bottom = max(j - winsz, 1), top = min(j + winsz, d.nz);
    float sn = 0;
    for (int s = bottom; s <= top; ++s) sn = sn + nsq[d.idx(vi - 1, s - 1, uk - 1)];
    const float curnsq = sn / (float)(top - bottom + 1);
    int npos = 1, nextn;
    for (int s = 1; s <= o.nn; ++s) if (curnsq > o.nsqv[s - 1]) npos = s;
    const float nweight = lw_weight(o.nsqv, o.nn, npos, nextn, curnsq);
    const size_t t = (size_t)(i - 1) + (size_t)nxu * (k - 1);
    const int4 b = brk[t]; const float2 wgt = brw[t];
    const float dweight = wgt.x, sweight = wgt.y;
    const int dpos = b.x, nextd = b.y, spos = b.z, nexts = b.w, np0 = npos - 1, nn0 = nextn - 1;
    const float luf = o.linear_update_fraction, lc = o.linear_contribution;
#define COMBO(s, dd, n) ((size_t)(s) + (size_t)o.ns * ((size_t)(dd) + (size_t)o.nd * (size_t)(n)))
    if (k <= d.ny) {
        const size_t cells = (size_t)nxu * d.nz * d.ny, c = (size_t)(i - 1) + (size_t)nxu * ((size_t)(j - 1) + (size_t)d.nz * (k - 1));
        const float *L = ulut + c;
        const float wind_first = nweight * (dweight * L[COMBO(spos, dpos, np0) * cells] + (1 - dweight) * L[COMBO(spos, nextd, np0) * cells])
                               + (1 - nweight) * (dweight * L[COMBO(spos, dpos, nn0) * cells] + (1 - dweight) * L[COMBO(spos, nextd, nn0) * cells]);
        const float wind_second = nweight * (dweight * L[COMBO(nexts, dpos, np0) * cells] + (1 - dweight) * L[COMBO(nexts, nextd, np0) * cells])
                                + (1 - nweight) * (dweight * L[COMBO(nexts, dpos, nn0) * cells] + (1 - dweight) * L[COMBO(nexts, nextd, nn0) * cells]);
        const float p = upert[c] * (1 - luf) + luf * (sweight * wind_first + (1 - sweight) * wind_second);
        upert[c] = p;
        u3d[c] = u3d[c] + p * lc;
    }
    if (i <= d.nx) {
        const size_t cells = (size_t)d.nx * d.nz * (d.ny + 1), c = (size_t)d.idx(i - 1, j - 1, k - 1);
        const float *L = vlut + c;
        const float wind_first = nweight * (dweight * L[COMBO(spos, dpos, np0) * cells] + (1 - dweight) * L[COMBO(spos, nextd, np0) * cells])
                               + (1 - nweight) * (dweight * L[COMBO(spos, dpos, nn0) * cells] + (1 - dweight) * L[COMBO(spos, nextd, nn0) * cells]);
        const float wind_second = nweight * (dweight * L[COMBO(nexts, dpos, np0) * cells] + (1 - dweight) * L[COMBO(nexts, nextd, np0) * cells])
                                + (1 - nweight) * (dweight * L[COMBO(nexts, dpos, nn0) * cells] + (1 - dweight) * L[COMBO(nexts, nextd, nn0) * cells]);
        const float p = vpert[c] * (1 - luf) + luf * (sweight * wind_first + (1 - sweight) * wind_second);
        vpert[c] = p;
        v3d[c] = v3d[c] + p * lc;
    }
#undef COMBO
}

__global__ void k_lw_exp(size_t n, float *__restrict__ x)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) x[t] = w_expf(x[t]);
}

// spatial_winds(domain, reverse=.false., vsmooth, winsz, update) :840-1127
int icar_spatial_winds_run(icar_hip_ctx *c, int update)
{
    LinWinds *w = c->linwinds;
    if (!w || !w->lut_ready) { icar_set_error("spatial_winds: the LUT has not been built or uploaded"); return 1; }
    if (!update) icar_winds_changed(c);                          // u, v change: the Courant winds of setup_winds are stale
    ScopedTimer t(c, "spatial_winds");
    const Dims d = c->d;
    float *u3d = update ? c->dqdt[ICAR_F_U] : icar_field_f(c, ICAR_F_U);
    float *v3d = update ? c->dqdt[ICAR_F_V] : icar_field_f(c, ICAR_F_V);
    if (!u3d || !v3d) { icar_set_error("spatial_winds: u/v (or their dqdt_3d when update) are not on the device"); return 1; }
    float *nsq = icar_field_f(c, ICAR_F_NSQUARED, false);
    if (!nsq) return 1;
    const icar_hip_lt_options &lo = w->o;
    LtDev o;
    o.variable_N = lo.variable_N; o.smooth_nsq = lo.smooth_nsq; o.N_squared = lo.N_squared;
    o.max_stability = lo.max_stability; o.min_stability = lo.min_stability;
    o.linear_contribution = lo.linear_contribution; o.linear_update_fraction = lo.linear_update_fraction;
    o.nd = lo.n_dir_values; o.ns = lo.n_spd_values; o.nn = lo.n_nsq_values;
    o.dirv = w->d_vals; o.spdv = w->d_vals + o.nd; o.nsqv = w->d_vals + o.nd + o.ns;
    const int vsmooth = lo.vert_smooth, winsz = lo.stability_window_size;
    if (lo.smooth_nsq && d.nx <= winsz) { icar_set_error("spatial_winds: smooth_array needs nx > stability_window_size"); return 1; }
    const float *th = nullptr, *ex = nullptr, *zz = nullptr, *qv = nullptr;
    if (lo.variable_N) {
        th = icar_field_f(c, ICAR_F_POTENTIAL_TEMPERATURE); ex = icar_field_f(c, ICAR_F_EXNER);
        zz = icar_field_f(c, ICAR_F_Z); qv = icar_field_f(c, ICAR_F_WATER_VAPOR);
        if (!th || !ex || !zz || !qv) return 1;
    }
    const float *qc = (const float *)c->field[ICAR_F_CLOUD_WATER], *qi = (const float *)c->field[ICAR_F_CLOUD_ICE];
    const float *qr = (const float *)c->field[ICAR_F_RAIN], *qs = (const float *)c->field[ICAR_F_SNOW];
    const dim3 b3(64, 4, 1);
    k_lw_nsq<<<dim3((d.nx + 63) / 64, (d.nz + 3) / 4, d.ny), b3, 0, c->stream>>>(o, d, vsmooth, th, ex, zz, qv, qc, qi, qr, qs, nsq);
    if (lo.smooth_nsq) {
        k_lw_vsmooth<<<dim3((d.nx + 63) / 64, d.ny), 64, 0, c->stream>>>(d, vsmooth, nsq);
        if (!w->rowmeans) HIPCHK(hipMalloc(&w->rowmeans, c->n3 * sizeof(double)));
        k_lw_rowmeans<<<dim3((d.nx + 63) / 64, d.nz), 64, 0, c->stream>>>(d, winsz, nsq, w->rowmeans);
        k_lw_colsmooth<<<dim3((d.nz + 63) / 64, d.ny), 64, 0, c->stream>>>(d, winsz, w->rowmeans, nsq);
    }
    const int nxu = d.nx + 1, nyv = d.ny + 1;
    if (!w->u1d) {
        const size_t n2 = (size_t)nxu * nyv;
        HIPCHK(hipMalloc(&w->u1d, n2 * sizeof(float))); HIPCHK(hipMalloc(&w->v1d, n2 * sizeof(float)));
        HIPCHK(hipMalloc(&w->brk, n2 * sizeof(int4))); HIPCHK(hipMalloc(&w->brw, n2 * sizeof(float2)));
    }
    // rows 1..ny read winds not yet modified in their own row; row ny+1 (v only) reads u(:,:,ny) AFTER its update,
    // exactly as the sequential reference loop does (:994-1001 with uk = min(k,ny)).
    const int ranges[2][2] = {{1, d.ny}, {nyv, nyv}};
    for (auto &r : ranges) {
        const int rows = r[1] - r[0] + 1;
        k_lw_means<<<dim3((nxu + 63) / 64, rows), 64, 0, c->stream>>>(o, d, r[0], u3d, v3d, w->u1d, w->v1d, w->brk, w->brw);
        k_lw_interp<<<dim3((nxu + 63) / 64, (d.nz + 3) / 4, rows), b3, 0, c->stream>>>(o, d, r[0], winsz, nsq, w->brk, w->brw, w->lut[0], w->lut[1],
                                                                                       w->pert[0], w->pert[1], u3d, v3d);
    }
    k_lw_exp<<<(unsigned)((c->n3 + 255) / 256), 256, 0, c->stream>>>(c->n3, nsq);
    HIPCHK(hipGetLastError());
    return 0;
}
