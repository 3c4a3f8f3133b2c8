// icar_amd/csrc/capi.hip -- extern "C" boundary (include/icar_hip.h), context and field registry,
// plus the small streaming kernels of rows H1 (halo faces), T2 (CFL reduction), W1 (balance_uvw).
#include "ctx.h"
#include "comm.h"
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>

static thread_local std::string g_err;
void icar_set_error(const std::string &msg) { g_err = msg; }
int icar_hip_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}

// ------------------------------------------------------------------------------------------------
// field registry
// ------------------------------------------------------------------------------------------------
static bool field_is_2dd(int f) { return f == ICAR_F_PRECIPITATION || f == ICAR_F_SNOWFALL || f == ICAR_F_GRAUPEL_ACC || f == ICAR_F_SINTHETA || f == ICAR_F_COSTHETA; }

size_t icar_field_count(const icar_hip_ctx *c, int f)
{
    const size_t nx = c->d.nx, nz = c->d.nz, ny = c->d.ny;
    if (f == ICAR_F_U || f == ICAR_F_JACOBIAN_U || f == ICAR_F_DZDX || f == ICAR_F_ZR_U) return (nx + 1) * nz * ny;
    if (f == ICAR_F_V || f == ICAR_F_JACOBIAN_V || f == ICAR_F_DZDY || f == ICAR_F_ZR_V) return nx * nz * (ny + 1);
    if (field_is_2dd(f) || f == ICAR_F_SURFACE_PRESSURE || (f >= ICAR_F_IVT && f <= ICAR_F_IWI)) return nx * ny;
    return nx * nz * ny;
}

float *icar_field_f(icar_hip_ctx *c, int f, bool required)
{
    if (f < 0 || f >= ICAR_N_FIELDS) { icar_set_error("bad field id"); return nullptr; }
    if (!c->field[f]) {
        if (required) {
            char b[96]; snprintf(b, sizeof b, "field %d has not been uploaded to the device", f);
            icar_set_error(b); return nullptr;
        }
        const size_t bytes = icar_field_count(c, f) * icar_hip_field_elem_size(f);
        if (icar_hip_check(hipMalloc(&c->field[f], bytes), "hipMalloc(field)")) return nullptr;
        if (icar_hip_check(hipMemsetAsync(c->field[f], 0, bytes, c->stream), "hipMemset(field)")) return nullptr;
    }
    return (float *)c->field[f];
}

ScopedTimer::ScopedTimer(icar_hip_ctx *c_, const char *g) : c(c_), group(g)
{
    if (!c->timing) return;
    // a timed event pair costs the stream ~5 us (two barrier packets with timestamps); a sub-step of a small tile has a dozen
    // groups: icar_hip_timing_groups restricts the timers to the ones that are read
    if (!c->timing_only.empty() && c->timing_only.find(std::string(",") + g + ",") == std::string::npos) return;
    auto take = [&](hipEvent_t &e) { if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else hipEventCreate(&e); };
    take(e0); take(e1);
    hipEventRecord(e0, c->stream);
}
ScopedTimer::~ScopedTimer()
{
    if (!e0) return;
    hipEventRecord(e1, c->stream);
    c->pending.push_back({group, {e0, e1}});
}

static void drain_timers(icar_hip_ctx *c)
{
    for (auto &p : c->pending) {
        float ms = 0;
        hipEventSynchronize(p.second.second);
        hipEventElapsedTime(&ms, p.second.first, p.second.second);
        auto &t = c->timers[p.first];
        t.total_ms += ms; t.launches += 1;
        c->event_pool.push_back(p.second.first); c->event_pool.push_back(p.second.second);
    }
    c->pending.clear();
}

// ------------------------------------------------------------------------------------------------
// H1: halo faces.  Buffer layout per field: N/S = [h][nz][nx] (verbatim planes), E/W = [ny][nz][h].
// ------------------------------------------------------------------------------------------------
struct HaloArgs { float *f[ICAR_MAX_ADV]; };

template <bool UNPACK>
__global__ void k_halo_ns(Dims d, HaloArgs a, int nv, int row0, int h, float *__restrict__ buf)
{
    // one thread per element of the h*nz*nx slab; i fastest => fully coalesced on both sides
    const size_t per = (size_t)d.nx * d.nz * h;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per) return;
    const int m = blockIdx.y;
    const size_t src = (size_t)row0 * d.sj + t;      // rows row0..row0+h-1 are contiguous in memory
    if (UNPACK) a.f[m][src] = buf[(size_t)m * per + t];
    else buf[(size_t)m * per + t] = a.f[m][src];
}

template <bool UNPACK>
__global__ void k_halo_ew(Dims d, HaloArgs a, int nv, int col0, int h, float *__restrict__ buf)
{
    // stride-nx gather: thread t -> (x = t % h, line = t / h), line = k + nz*j
    const size_t per = (size_t)h * d.nz * d.ny;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per) return;
    const int m = blockIdx.y;
    const int x = (int)(t % h);
    const size_t line = t / h;
    const size_t src = line * d.nx + col0 + x;
    if (UNPACK) a.f[m][src] = buf[(size_t)m * per + t];
    else buf[(size_t)m * per + t] = a.f[m][src];
}

int icar_halo_pack(icar_hip_ctx *c, int dir, int h, const int *fields, int n, float *buf, bool unpack)
{
    if (n <= 0) return 0;
    if (n > ICAR_MAX_ADV) { icar_set_error("halo: too many fields"); return 1; }
    if (h < 1 || 2 * h > c->d.nx || 2 * h > c->d.ny) { icar_set_error("halo: bad halo width"); return 1; }
    HaloArgs a;
    for (int m = 0; m < n; ++m) {
        if (fields[m] < 0 || fields[m] >= ICAR_N_ADVECTABLE) { icar_set_error("halo: only exchangeable scalars"); return 1; }
        a.f[m] = icar_field_f(c, fields[m]);
        if (!a.f[m]) return 1;
    }
    const Dims &d = c->d;
    ScopedTimer t(c, "halo");
    if (dir == 0 || dir == 1) {
        // put_north sends rows ny-2h..ny-h-1 ; put_south rows h..2h-1          (exchangeable_obj.f90:263,280)
        // retrieve_north fills rows ny-h..ny-1 ; retrieve_south rows 0..h-1    (:290,:300)
        int row0;
        if (!unpack) row0 = (dir == 0) ? d.ny - 2 * h : h;
        else         row0 = (dir == 0) ? d.ny - h : 0;
        const size_t per = (size_t)d.nx * d.nz * h;
        dim3 g((unsigned)((per + 255) / 256), n), b(256);
        if (unpack) hipLaunchKernelGGL(k_halo_ns<true>, g, b, 0, c->stream, d, a, n, row0, h, buf);
        else        hipLaunchKernelGGL(k_halo_ns<false>, g, b, 0, c->stream, d, a, n, row0, h, buf);
    } else if (dir == 2 || dir == 3) {
        // put_east sends cols nx-2h..nx-h-1 ; put_west cols h..2h-1            (:317,:335)
        int col0;
        if (!unpack) col0 = (dir == 2) ? d.nx - 2 * h : h;
        else         col0 = (dir == 2) ? d.nx - h : 0;
        const size_t per = (size_t)h * d.nz * d.ny;
        dim3 g((unsigned)((per + 255) / 256), n), b(256);
        if (unpack) hipLaunchKernelGGL(k_halo_ew<true>, g, b, 0, c->stream, d, a, n, col0, h, buf);
        else        hipLaunchKernelGGL(k_halo_ew<false>, g, b, 0, c->stream, d, a, n, col0, h, buf);
    } else { icar_set_error("halo: dir must be 0..3"); return 1; }
    HIPCHK(hipGetLastError());
    return 0;
}

// all directions of one halo_send / halo_retrieve in ONE launch (blockIdx.z = direction): a step of an image with four
// neighbours issues 2 launches instead of 8 -- small tiles are host-launch-bound
struct HaloDirs { int n, ns[4], start[4], skip_w, skip_e; float *buf[4]; };
template <bool UNPACK>
__global__ void k_halo_dirs(Dims d, HaloArgs a, int h, HaloDirs hd)
{
    const int z = blockIdx.z, m = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float *__restrict__ buf = hd.buf[z];
    if (hd.ns[z]) {                                                // N/S: h contiguous rows
        const size_t per = (size_t)d.nx * d.nz * h;
        if (t >= per) return;
        const size_t src = (size_t)hd.start[z] * d.sj + t;
        if (UNPACK) {
            // the corner cells belong to both an N/S row and an E/W column; the reference retrieves N, S, E, W in that
            // order (exchangeable_obj.f90:138-151), so E/W win: with them in the same launch the rows leave the corners alone
            const int i = (int)(t % d.nx);
            if ((hd.skip_w && i < h) || (hd.skip_e && i >= d.nx - h)) return;
            a.f[m][src] = buf[(size_t)m * per + t];
        } else buf[(size_t)m * per + t] = a.f[m][src];
    } else {                                                       // E/W: h columns of every (k, j) line
        const size_t per = (size_t)h * d.nz * d.ny;
        if (t >= per) return;
        const int x = (int)(t % h); const size_t line = t / h;
        const size_t src = line * d.nx + hd.start[z] + x;
        if (UNPACK) a.f[m][src] = buf[(size_t)m * per + t]; else buf[(size_t)m * per + t] = a.f[m][src];
    }
}

int icar_halo_pack_dirs(icar_hip_ctx *c, int ndir, const int *dirs, int h, const int *fields, int n, void *const *bufs, bool unpack)
{
    if (n <= 0 || ndir <= 0) return 0;
    if (ndir > 4) { icar_set_error("halo: at most 4 directions per call"); return 1; }
    if (n > ICAR_MAX_ADV) { icar_set_error("halo: too many fields"); return 1; }
    if (h < 1 || 2 * h > c->d.nx || 2 * h > c->d.ny) { icar_set_error("halo: bad halo width"); return 1; }
    HaloArgs a;
    for (int m = 0; m < n; ++m) {
        if (fields[m] < 0 || fields[m] >= ICAR_N_ADVECTABLE) { icar_set_error("halo: only exchangeable scalars"); return 1; }
        a.f[m] = icar_field_f(c, fields[m]);
        if (!a.f[m]) return 1;
    }
    const Dims &d = c->d;
    HaloDirs hd; hd.n = ndir; hd.skip_w = hd.skip_e = 0;
    size_t permax = 0;
    for (int z = 0; z < ndir; ++z) {
        const int dir = dirs[z];
        if (dir < 0 || dir > 3 || !bufs[z]) { icar_set_error("halo: dir must be 0..3 with a buffer"); return 1; }
        hd.ns[z] = (dir < 2); hd.buf[z] = (float *)bufs[z];
        if (dir == 2) hd.skip_e = 1;
        if (dir == 3) hd.skip_w = 1;
        // same planes as icar_halo_pack: put_north rows ny-2h.., put_south rows h.., put_east cols nx-2h.., put_west cols h.. ;
        // retrieve fills the outermost h rows / columns
        if (dir < 2) hd.start[z] = !unpack ? (dir == 0 ? d.ny - 2 * h : h) : (dir == 0 ? d.ny - h : 0);
        else         hd.start[z] = !unpack ? (dir == 2 ? d.nx - 2 * h : h) : (dir == 2 ? d.nx - h : 0);
        permax = std::max(permax, dir < 2 ? (size_t)d.nx * d.nz * h : (size_t)h * d.nz * d.ny);
    }
    ScopedTimer t(c, "halo");
    dim3 g((unsigned)((permax + 255) / 256), n, ndir), b(256);
    if (unpack) hipLaunchKernelGGL(k_halo_dirs<true>, g, b, 0, c->stream, d, a, h, hd);
    else        hipLaunchKernelGGL(k_halo_dirs<false>, g, b, 0, c->stream, d, a, h, hd);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// T2: compute_dt strictness-3 reduction (time_step.f90:264-289)
// ------------------------------------------------------------------------------------------------
// One thread per (i, j) column, levels marched in registers (|w| of the level below is carried), raw buffer loads with scalar
// row / level offsets: at most 16 VGPRs (the attribute counts half of the unified file), so that the prefetched reduction
// finds a wave slot on EVERY SIMD beside the MPDATA launch it is issued next to (whose persistent blocks leave 16 registers per
// SIMD; the former grid-stride kernel needed 61 and ran on the 13 idle CUs only: 0.38 ms in the advection's shadow).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(8)))
k_max_courant(Dims d, const float *__restrict__ u, const float *__restrict__ v,
              const float *__restrict__ w, const float *__restrict__ dzl, float dx,
              unsigned *__restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto mk = [](const float *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, -1, 0x00020000); };
    const rsrc_t ru = mk(u), rv = mk(v), rw = mk(w);
    auto ld = [](rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
    float cur = 0.0f;
    if (i < d.nx) {
        const int vi = 4 * i, nxu = d.nx + 1;
        float wbelow = 0.0f;
#pragma unroll 1
        for (int k = 0; k < d.nz; ++k) {
            const int sc = 4 * d.idx(0, k, j), scu = 4 * (nxu * (k + d.nz * j));     // wave-uniform
            const float au = fmaxf(fabsf(ld(ru, vi, scu)), fabsf(ld(ru, vi, scu + 4)));
            const float av = fmaxf(fabsf(ld(rv, vi, sc)), fabsf(ld(rv, vi, sc + 4 * d.sj)));
            const float aw0 = fabsf(ld(rw, vi, sc));
            const float aw = (k == 0) ? aw0 : fmaxf(aw0, wbelow);                     // (level 0 looks at itself, :281)
            const float cw = au / dx + av / dx + aw / dzl[k];
            cur = fmaxf(cur, cw);
            wbelow = aw0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cur = fmaxf(cur, __shfl_down(cur, o));
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(cur));   // non-negative floats order like unsigned ints
}

// out != nullptr: the maximum is copied to the host (one stream synchronisation).  d_out != nullptr: it is left in device
// memory at d_out (a REAL(4) the caller owns) and nothing waits -- the caller all-reduces it on the device (co_min of
// time_step.f90:413 as max over images of the Courant sum: dt = factor / max is monotone, so min(dt) == factor / max).
// A maximum taken ahead of time (icar_hip_max_courant_prefetch, typically on the second stream beside the advection) is
// handed out instead of a new reduction as long as no entry point has written u, v or w since and the arguments are the same.
static bool cfl_prefetched(icar_hip_ctx *c, float dx, const float *dz_levels)
{
    return c->cfl_pre.valid && !c->wind_ptr_escaped && c->cfl_pre.ver == c->wind_version && c->cfl_pre.dx == dx
        && (int)c->cfl_pre.dzl.size() == c->d.nz && memcmp(c->cfl_pre.dzl.data(), dz_levels, sizeof(float) * c->d.nz) == 0;
}

int icar_max_courant_run(icar_hip_ctx *c, float dx, const float *dz_levels, float *out, float *d_out)
{
    if (cfl_prefetched(c, dx, dz_levels) && !c->cfl_pre.reduced) {      // (a value already reduced over the images is not this tile's)
        c->cfl_pre.valid = false;
        if (out) { HIPCHK(hipEventSynchronize(c->cfl_ev)); *out = *c->h_cfl_pre; }
        else {
            HIPCHK(hipStreamWaitEvent(c->stream, c->cfl_ev, 0));
            HIPCHK(hipMemcpyAsync(d_out, c->d_red + 8, sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        return 0;
    }
    c->cfl_pre.valid = false;
    const float *u = icar_field_f(c, ICAR_F_U), *v = icar_field_f(c, ICAR_F_V), *w = icar_field_f(c, ICAR_F_W);
    if (!u || !v || !w) return 1;
    float *dzl = c->d_red + 16;
    if ((int)c->dzl_host.size() != c->d.nz || memcmp(c->dzl_host.data(), dz_levels, sizeof(float) * c->d.nz) != 0) {
        c->dzl_host.assign(dz_levels, dz_levels + c->d.nz);      // the copy source must outlive the async copy
        HIPCHK(hipMemcpyAsync(dzl, c->dzl_host.data(), sizeof(float) * c->d.nz, hipMemcpyHostToDevice, c->stream));
    }
    float *red = d_out ? d_out : c->d_red;
    HIPCHK(hipMemsetAsync(red, 0, sizeof(float), c->stream));
    if ((size_t)(c->d.nx + 1) * c->d.nz * (c->d.ny + 1) * sizeof(float) >= ((size_t)1 << 31)) { icar_set_error("max_courant: a field of 2 GiB or more is not supported (32-bit buffer offsets)"); return 1; }
    dim3 g((c->d.nx + 63) / 64, c->d.ny), b(64);
    ScopedTimer t(c, "cfl");
    hipLaunchKernelGGL(k_max_courant, g, b, 0, c->stream, c->d, u, v, w, dzl, dx, (unsigned *)red);
    HIPCHK(hipGetLastError());
    if (out) {
        HIPCHK(hipMemcpyAsync(out, red, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// the prefetched maximum, if it is still valid AND was already reduced over the images on the device (allreduce = true below):
// consumes it; the host waits for the second stream's copy only
bool icar_cfl_prefetch_waiting(icar_hip_ctx *c)
{
    return c->step.configured && cfl_prefetched(c, c->step.cfg.dx, c->step.dz_levels.data());
}

bool icar_cfl_prefetched_global(icar_hip_ctx *c, float dx, const float *dz_levels, float *value)
{
    if (!cfl_prefetched(c, dx, dz_levels) || !c->cfl_pre.reduced) return false;
    c->cfl_pre.valid = false;
    // (a failed wait is an error of this image alone; falling back to a fresh reduction here would issue an all-reduce the other
    // images do not pair -- the value is handed out as NaN and compute_dt reports it)
    if (hipEventSynchronize(c->cfl_ev) != hipSuccess) { *value = __builtin_nanf(""); c->cfl_wait_failed = true; return true; }
    *value = *c->h_cfl_pre;
    return true;
}

// allreduce: with the RCCL transport the tile maximum is all-reduced (MAX) over the images right here, on the current (second)
// stream in the advection's shadow, so that the next update_dt finds the GLOBAL maximum waiting instead of paying an
// all-reduce + two copies on the critical path.  Every image takes the same decisions (SPMD), so the collective calls pair.
int icar_max_courant_prefetch_run(icar_hip_ctx *c, float dx, const float *dz_levels, bool allreduce)
{
    c->cfl_pre.valid = false;
    if (!c->h_cfl_pre) { HIPCHK(hipHostMalloc((void **)&c->h_cfl_pre, sizeof(float), hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&c->cfl_ev, hipEventDisableTiming)); }
    if (icar_max_courant_run(c, dx, dz_levels, nullptr, c->d_red + 8)) return 1;             // on the current stream, nothing waits
    if (allreduce && icar_comm_max_device(c, c->d_red + 8) != 0) return 1;
    HIPCHK(hipMemcpyAsync(c->h_cfl_pre, c->d_red + 8, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->cfl_ev, c->stream));
    c->cfl_pre.reduced = allreduce;
    c->cfl_pre.valid = true; c->cfl_pre.ver = c->wind_version; c->cfl_pre.dx = dx; c->cfl_pre.dzl.assign(dz_levels, dz_levels + c->d.nz);
    return 0;
}

// maxval(abs(u)), maxval(abs(v)), maxval(abs(w)) of the other cfl_strictness settings (time_step.f90:238-259, :293-305)
__global__ void __launch_bounds__(256)
k_max_abs3(size_t nu, size_t nv, size_t nw, const float *__restrict__ u, const float *__restrict__ v,
           const float *__restrict__ w, unsigned *__restrict__ out)
{
    const float *x = blockIdx.y == 0 ? u : blockIdx.y == 1 ? v : w;
    const size_t n = blockIdx.y == 0 ? nu : blockIdx.y == 1 ? nv : nw;
    float cur = 0.0f;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) cur = fmaxf(cur, fabsf(x[t]));
    for (int o = 32; o > 0; o >>= 1) cur = fmaxf(cur, __shfl_down(cur, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out + blockIdx.y, __float_as_uint(cur));
}

int icar_max_abs_winds_run(icar_hip_ctx *c, float *out3)
{
    const float *u = icar_field_f(c, ICAR_F_U), *v = icar_field_f(c, ICAR_F_V), *w = icar_field_f(c, ICAR_F_W);
    if (!u || !v || !w) return 1;
    HIPCHK(hipMemsetAsync(c->d_red, 0, 3 * sizeof(float), c->stream));
    ScopedTimer t(c, "cfl");
    hipLaunchKernelGGL(k_max_abs3, dim3(512, 3), dim3(256), 0, c->stream, icar_field_count(c, ICAR_F_U), icar_field_count(c, ICAR_F_V),
                       icar_field_count(c, ICAR_F_W), u, v, w, (unsigned *)c->d_red);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out3, c->d_red, 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// W1: balance_uvw (wind.f90:81-169): w from the horizontal divergence, bottom-up per column
// ------------------------------------------------------------------------------------------------
__global__ void k_balance_uvw(Dims d, const float *__restrict__ u, const float *__restrict__ v, float *__restrict__ w,
                              const float *__restrict__ ju, const float *__restrict__ jv, const float *__restrict__ jw,
                              const float *__restrict__ dz, float dx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= d.nx) return;
    float wprev = 0.0f, jwprev = 0.0f;
    for (int k = 0; k < d.nz; ++k) {
        const int c = d.idx(i, k, j);
        const int cu = i + (d.nx + 1) * (k + d.nz * j);
        const float du = u[cu + 1] * ju[cu + 1] - u[cu] * ju[cu];       // calc_divergence :207-210
        const float dv = v[c + d.sj] * jv[c + d.sj] - v[c] * jv[c];
        const float div = (du + dv) / dx;
        float wk;
        if (k == 0) wk = 0 - div * dz[c] / jw[c];                       // :141
        else        wk = (wprev * jwprev - div * dz[c]) / jw[c];        // :143
        w[c] = wk; wprev = wk; jwprev = jw[c];
    }
}

int icar_balance_uvw_run(icar_hip_ctx *c, float dx, int update)
{
    // update != 0: wind.f90:341-360 balances the forcing tendencies u/v/w%meta_data%dqdt_3d instead of the winds
    const float *u = update ? c->dqdt[ICAR_F_U] : icar_field_f(c, ICAR_F_U), *v = update ? c->dqdt[ICAR_F_V] : icar_field_f(c, ICAR_F_V);
    const float *ju = icar_field_f(c, ICAR_F_JACOBIAN_U), *jv = icar_field_f(c, ICAR_F_JACOBIAN_V);
    const float *jw = icar_field_f(c, ICAR_F_JACOBIAN_W), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    float *w = nullptr;
    if (update) {
        if (!c->dqdt[ICAR_F_W]) {
            if (icar_hip_check(hipMalloc(&c->dqdt[ICAR_F_W], c->n3 * sizeof(float)), "hipMalloc(dqdt w)")) return 1;
        }
        w = c->dqdt[ICAR_F_W];
        if (!u || !v) { icar_set_error("balance_uvw(update): upload the u and v dqdt_3d first (icar_hip_dqdt_upload)"); return 1; }
    } else w = icar_field_f(c, ICAR_F_W, false);
    if (!u || !v || !ju || !jv || !jw || !dz || !w) return 1;
    dim3 g((c->d.nx + 63) / 64, c->d.ny), b(64);
    hipLaunchKernelGGL(k_balance_uvw, g, b, 0, c->stream, c->d, u, v, w, ju, jv, jw, dz, dx);
    HIPCHK(hipGetLastError());
    if (!update) icar_winds_changed(c);                          // w changed: the Courant winds are stale
    return 0;
}

// ------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *icar_hip_last_error(void) { return g_err.c_str(); }
const char *icar_hip_version(void) { return "icar_hip 0.1 (gfx950)"; }
size_t icar_hip_field_elem_size(int field) { return field_is_2dd(field) ? sizeof(double) : sizeof(float); }
size_t icar_hip_field_count(const icar_hip_ctx *ctx, int field) { return ctx ? icar_field_count(ctx, field) : 0; }

int icar_hip_ctx_create(icar_hip_ctx **out, int device, int ims, int ime, int kms, int kme, int jms, int jme)
{
    if (!out) { icar_set_error("ctx_create: null out"); return 1; }
    *out = nullptr;
    if (ime < ims + 2 || jme < jms + 2 || kme < kms + 1) { icar_set_error("ctx_create: tile must be at least 3x2x3"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        icar_set_error("ctx_create: no HIP device visible (libicar_hip has no CPU fallback)"); return 1;
    }
    if (device < 0 || device >= ndev) { icar_set_error("ctx_create: bad device index"); return 1; }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950")) {
        icar_set_error(std::string("ctx_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return 1;
    }
    icar_hip_ctx *c = new icar_hip_ctx();
    c->device = device;
    c->ims = ims; c->ime = ime; c->kms = kms; c->kme = kme; c->jms = jms; c->jme = jme;
    c->d.nx = ime - ims + 1; c->d.nz = kme - kms + 1; c->d.ny = jme - jms + 1;
    c->d.sk = c->d.nx; c->d.sj = c->d.nx * c->d.nz;
    c->n3 = (size_t)c->d.nx * c->d.nz * c->d.ny;
    if ((size_t)(c->d.nx + 1) * c->d.nz * (c->d.ny + 1) >= (size_t)1 << 31) { delete c; icar_set_error("ctx_create: tile too large for 32-bit indexing"); return 1; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; icar_set_error("hipStreamCreate failed"); return 1; }
    c->own_stream = true;
    if (hipMalloc(&c->d_red, sizeof(float) * (16 + 4096)) != hipSuccess || hipMalloc(&c->d_flag, sizeof(int) * 16) != hipSuccess) {
        delete c; icar_set_error("hipMalloc failed"); return 1;
    }
    *out = c;
    return 0;
}

int icar_hip_ctx_destroy(icar_hip_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (int f = 0; f < ICAR_N_FIELDS; ++f) if (c->field[f]) hipFree(c->field[f]);
    for (int f = 0; f < ICAR_N_ADVECTABLE; ++f) if (c->alt[f]) hipFree(c->alt[f]);
    for (int f = 0; f < ICAR_N_FIELDS; ++f) if (c->dqdt[f]) hipFree(c->dqdt[f]);
    float *scr[] = {c->U, c->V, c->W, c->Wdz, c->d_red, c->mpc, c->mpx_buf};
    for (float *p : scr) if (p) hipFree(p);
    if (c->d_flag) hipFree(c->d_flag);
    if (c->h_cfl_pre) hipHostFree(c->h_cfl_pre);
    if (c->cfl_ev) hipEventDestroy(c->cfl_ev);
    if (c->iw_adj) hipFree(c->iw_adj);
    if (c->wgr_tmp) hipFree(c->wgr_tmp);
    icar_wsm3_free(c);
    icar_wsm6_free(c);
    icar_thompson_free(c);
    icar_linwinds_free(c);
    icar_comm_free(c);
    if (c->step.h_val) hipHostFree(c->step.h_val);
    if (c->on_aux) c->stream = c->main_saved;
    if (c->aux) { hipStreamSynchronize(c->aux); hipStreamDestroy(c->aux); }
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int icar_hip_set_stream(icar_hip_ctx *c, void *s)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (c->on_aux) { icar_set_error("set_stream: called between aux_begin and aux_end"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s) {
        if (c->own_stream) { hipStreamDestroy(c->stream); c->own_stream = false; }
        c->stream = (hipStream_t)s;
    } else if (!c->own_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return 0;
}

int icar_hip_synchronize(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_hip_field_upload(icar_hip_ctx *c, int f, const void *host)
{
    if (!c || !host) { icar_set_error("field_upload: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    float *p = icar_field_f(c, f, false);
    if (!p) return 1;
    HIPCHK(hipMemcpyAsync(p, host, icar_field_count(c, f) * icar_hip_field_elem_size(f), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (f == ICAR_F_U || f == ICAR_F_V || f == ICAR_F_W || (f >= ICAR_F_DENSITY && f <= ICAR_F_ADVECTION_DZ)) icar_winds_changed(c);
    return 0;
}

int icar_hip_field_download(icar_hip_ctx *c, int f, void *host)
{
    if (!c || !host) { icar_set_error("field_download: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    float *p = icar_field_f(c, f, true);
    if (!p) return 1;
    HIPCHK(hipMemcpyAsync(host, p, icar_field_count(c, f) * icar_hip_field_elem_size(f), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

__global__ void k_fill_f(float *p, size_t n, float v) { size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (t < n) p[t] = v; }
__global__ void k_fill_d(double *p, size_t n, double v) { size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (t < n) p[t] = v; }

int icar_hip_field_fill(icar_hip_ctx *c, int f, double value)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    float *p = icar_field_f(c, f, false);
    if (!p) return 1;
    const size_t n = icar_field_count(c, f);
    if (field_is_2dd(f)) hipLaunchKernelGGL(k_fill_d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (double *)p, n, value);
    else                 hipLaunchKernelGGL(k_fill_f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, p, n, (float)value);
    HIPCHK(hipGetLastError());
    if (f == ICAR_F_U || f == ICAR_F_V || f == ICAR_F_W || (f >= ICAR_F_DENSITY && f <= ICAR_F_ADVECTION_DZ)) icar_winds_changed(c);
    return 0;
}

int icar_hip_field_device_ptr(icar_hip_ctx *c, int f, void **dptr)
{
    if (!c || !dptr) { icar_set_error("field_device_ptr: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    float *p = icar_field_f(c, f, false);
    if (!p) return 1;
    *dptr = p;
    if (f == ICAR_F_U || f == ICAR_F_V || f == ICAR_F_W) { c->wind_ptr_escaped = true; icar_winds_changed(c); }   // the caller may write them behind our back
    return 0;
}

int icar_hip_setup_winds(icar_hip_ctx *c, int scheme, float dt, float dx, int advect_density)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_advect_setup_winds(c, scheme, dt, dx, advect_density);
}

int icar_hip_advect(icar_hip_ctx *c, int scheme, int mpdata_order, int fct, int advect_density, const int *fields, int nfields)
{
    if (!c || (!fields && nfields > 0)) { icar_set_error("advect: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_advect_run(c, scheme, mpdata_order, fct, advect_density, fields, nfields);
}

int icar_hip_mpdata_exact(icar_hip_ctx *c, int on)
{
    if (!c || (on != 0 && on != 1)) { icar_set_error("mpdata_exact: ctx and on = 0 / 1"); return 1; }
    c->mpdata_exact = on;
    return 0;
}

int icar_hip_mp_simple(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte, int *err_count)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_mp_simple_run(c, dt, its, ite, jts, jte, kts, kte, err_count);
}

int icar_hip_mp_simple_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte, int *err_count)
{
    if (!c || !tiles) { icar_set_error("mp_simple_tiles: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_mp_simple_run_tiles(c, dt, ntiles, tiles, kts, kte, err_count);
}

int icar_hip_thompson_init(icar_hip_ctx *c, const float params[18], const int flags[2])
{
    if (!c || !params || !flags) { icar_set_error("thompson_init: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_thompson_init_run(c, params, flags);
}

int icar_hip_thompson(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte,
                      int ids, int ide, int jds, int jde, int kds, int kde)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_thompson_run(c, dt, its, ite, jts, jte, kts, kte, ids, ide, jds, jde, kds, kde);
}

int icar_hip_thompson_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte,
                            int ids, int ide, int jds, int jde, int kds, int kde)
{
    if (!c || !tiles) { icar_set_error("thompson_tiles: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_thompson_run_tiles(c, dt, ntiles, tiles, kts, kte, ids, ide, jds, jde, kds, kde);
}

int icar_hip_thompson_table(icar_hip_ctx *c, const char *name, double *out, size_t capacity, size_t *count)
{
    if (!c || !name) { icar_set_error("thompson_table: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_thompson_table_download(c, name, out, capacity, count);
}

int icar_hip_mp_tiles(int its, int ite, int jts, int jte, int halo, int subset, int tiles[4][4])
{
    // mp_driver.f90:609-658 (process_halo) and :728-737 (subset)
    if (halo > 0) {
        const int t[4][4] = {
            {its, its + halo - 1, jts, jte},                     // west strip, full height
            {ite - halo + 1, ite, jts, jte},                     // east strip
            {its + halo, ite - halo, jts, jts + halo - 1},       // south strip without corners
            {its + halo, ite - halo, jte - halo + 1, jte}};      // north strip
        memcpy(tiles, t, sizeof t);
        return 4;
    }
    tiles[0][0] = its + subset; tiles[0][1] = ite - subset; tiles[0][2] = jts + subset; tiles[0][3] = jte - subset;
    return 1;
}

int icar_hip_max_courant(icar_hip_ctx *c, float dx, const float *dz_levels, float *out)
{
    if (!c || !dz_levels || !out) { icar_set_error("max_courant: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (c->d.nz > 4096) { icar_set_error("max_courant: nz too large"); return 1; }
    return icar_max_courant_run(c, dx, dz_levels, out, nullptr);
}

int icar_hip_max_courant_device(icar_hip_ctx *c, float dx, const float *dz_levels, void *d_out)
{
    if (!c || !dz_levels || !d_out) { icar_set_error("max_courant_device: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (c->d.nz > 4096) { icar_set_error("max_courant_device: nz too large"); return 1; }
    return icar_max_courant_run(c, dx, dz_levels, nullptr, (float *)d_out);
}

int icar_hip_max_courant_prefetch(icar_hip_ctx *c, float dx, const float *dz_levels)
{
    if (!c || !dz_levels) { icar_set_error("max_courant_prefetch: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (c->d.nz > 4096) { icar_set_error("max_courant_prefetch: nz too large"); return 1; }
    return icar_max_courant_prefetch_run(c, dx, dz_levels, false);
}

int icar_hip_diagnostic_update(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_diagnostic_update_run(c, 3);
}

int icar_hip_diagnostic_update_parts(icar_hip_ctx *c, int parts)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (parts < 1 || parts > 3) { icar_set_error("diagnostic_update_parts: parts is 1 (thermodynamics), 2 (w_real) or 3"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_diagnostic_update_run(c, parts);
}

int icar_hip_dqdt_upload(icar_hip_ctx *c, int f, const void *host)
{
    if (!c || !host) { icar_set_error("dqdt_upload: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (f < 0 || f >= ICAR_N_FIELDS || field_is_2dd(f)) { icar_set_error("dqdt_upload: bad field"); return 1; }
    const size_t bytes = icar_field_count(c, f) * sizeof(float);
    if (!c->dqdt[f]) HIPCHK(hipMalloc(&c->dqdt[f], bytes));
    HIPCHK(hipMemcpyAsync(c->dqdt[f], host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_hip_apply_forcing(icar_hip_ctx *c, double dt, const int *fields, const int *fb, int n, int w, int e, int s, int nn)
{
    if (!c || (n > 0 && (!fields || !fb))) { icar_set_error("apply_forcing: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_apply_forcing_run(c, dt, fields, fb, n, w, e, s, nn);
}

int icar_hip_enforce_limits(icar_hip_ctx *c, const int *fields, int n)
{
    if (!c || (n > 0 && !fields)) { icar_set_error("enforce_limits: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_enforce_limits_run(c, fields, n);
}

int icar_hip_wsm6_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte)
{
    if (!c || !tiles) { icar_set_error("wsm6_tiles: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_wsm6_run_tiles(c, dt, ntiles, tiles, kts, kte);
}

int icar_hip_winds_valid(icar_hip_ctx *c) { return (c && c->winds_valid) ? 1 : 0; }

int icar_hip_wsm6_init(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_wsm6_init_run(c);
}

int icar_hip_wsm6(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_wsm6_run(c, dt, its, ite, jts, jte, kts, kte);
}

int icar_hip_wsm3_init(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_wsm3_init_run(c);
}

int icar_hip_wsm3(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_wsm3_run(c, dt, its, ite, jts, jte, kts, kte);
}

int icar_hip_max_abs_winds(icar_hip_ctx *c, float *out3)
{
    if (!c || !out3) { icar_set_error("max_abs_winds: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_max_abs_winds_run(c, out3);
}

int icar_hip_balance_uvw(icar_hip_ctx *c, float dx)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_balance_uvw_run(c, dx, 0);
}

int icar_hip_balance_uvw_update(icar_hip_ctx *c, float dx)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_balance_uvw_run(c, dx, 1);
}

int icar_hip_make_winds_grid_relative(icar_hip_ctx *c, int update)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_make_winds_grid_relative(c, update);
}

int icar_hip_mass_conservative_acceleration(icar_hip_ctx *c, int update)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_mass_conservative_acceleration(c, update);
}

int icar_hip_iterative_winds_correct_w(icar_hip_ctx *c, int update)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_iterative_winds_correct_w(c, update);
}

int icar_hip_iterative_winds_sweep(icar_hip_ctx *c, float dx, int nsweeps, int update)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (nsweeps < 0 || !(dx > 0)) { icar_set_error("iterative_winds_sweep: bad argument"); return 1; }
    return icar_iterative_winds_sweep(c, dx, nsweeps, update);
}

int icar_hip_box_pack(icar_hip_ctx *c, int field, int which, int i0, int ni, int j0, int nj, void *dbuf)
{
    if (!c || !dbuf) { icar_set_error("box_pack: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_box_copy(c, field, which, i0, ni, j0, nj, (float *)dbuf, false);
}

int icar_hip_box_unpack(icar_hip_ctx *c, int field, int which, int i0, int ni, int j0, int nj, const void *dbuf)
{
    if (!c || !dbuf) { icar_set_error("box_unpack: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (which == 0 && (field == ICAR_F_U || field == ICAR_F_V || field == ICAR_F_W)) icar_winds_changed(c);
    return icar_box_copy(c, field, which, i0, ni, j0, nj, (float *)dbuf, true);
}

int icar_hip_dqdt_download(icar_hip_ctx *c, int f, void *host)
{
    if (!c || !host) { icar_set_error("dqdt_download: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (f < 0 || f >= ICAR_N_FIELDS || field_is_2dd(f) || !c->dqdt[f]) { icar_set_error("dqdt_download: no dqdt mirror for this field"); return 1; }
    HIPCHK(hipMemcpyAsync(host, c->dqdt[f], icar_field_count(c, f) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int icar_hip_linwinds_setup(icar_hip_ctx *c, const icar_hip_lt_options *opt, const float *global_terrain,
                            int nx_global, int ny_global, int ids, int jds, float dx)
{
    if (!c || !opt || !global_terrain) { icar_set_error("linwinds_setup: null argument"); return 1; }
    if (nx_global < 2 || ny_global < 2 || !(dx > 0)) { icar_set_error("linwinds_setup: bad global terrain size / dx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_setup_run(c, opt, global_terrain, nx_global, ny_global, ids, jds, dx);
}

int icar_hip_linwinds_terrain_frequency(icar_hip_ctx *c, double *out, size_t cap, int *fftnx, int *fftny)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_terrain_frequency(c, out, cap, fftnx, fftny);
}

int icar_hip_linear_perturbation(icar_hip_ctx *c, float U, float V, float Nsq, float z_bottom, float z_top,
                                 float minimum_step, double *u_perturb, double *v_perturb)
{
    if (!c || !u_perturb || !v_perturb) { icar_set_error("linear_perturbation: null argument"); return 1; }
    if (!(minimum_step > 0) || !(z_top > z_bottom)) { icar_set_error("linear_perturbation: need z_top > z_bottom and minimum_step > 0"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linear_perturbation_run(c, U, V, Nsq, z_bottom, z_top, minimum_step, u_perturb, v_perturb);
}

int icar_hip_linwinds_build_lut(icar_hip_ctx *c, const float *z_bottom, const float *z_top, int nz)
{
    if (!c || !z_bottom || !z_top) { icar_set_error("linwinds_build_lut: null argument"); return 1; }
    for (int k = 0; k < nz; ++k) if (!(z_top[k] > z_bottom[k])) { icar_set_error("linwinds_build_lut: need z_top > z_bottom on every level"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_build_lut_run(c, z_bottom, z_top, nz);
}

int icar_hip_linwinds_build_lut_varying(icar_hip_ctx *c, const float *z_bottom, const float *z_top, int nz)
{
    if (!c || !z_bottom || !z_top) { icar_set_error("linwinds_build_lut_varying: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_build_lut_varying_run(c, z_bottom, z_top, nz);
}

int icar_hip_linwinds_lut_download(icar_hip_ctx *c, int comp, float *host)
{
    if (!c || !host) { icar_set_error("linwinds_lut_download: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_lut_copy(c, comp, host, 0);
}

int icar_hip_linwinds_lut_entry(icar_hip_ctx *c, int comp, int spd, int dir, int nsq, float *host)
{
    if (!c || !host) { icar_set_error("linwinds_lut_entry: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_lut_entry(c, comp, spd, dir, nsq, host);
}

int icar_hip_linwinds_lut_upload(icar_hip_ctx *c, int comp, const float *host)
{
    if (!c || !host) { icar_set_error("linwinds_lut_upload: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_lut_copy(c, comp, const_cast<float *>(host), 1);
}

int icar_hip_linwinds_perturbation_download(icar_hip_ctx *c, int comp, float *host)
{
    if (!c || !host) { icar_set_error("linwinds_perturbation_download: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_pert_copy(c, comp, host, 0);
}

int icar_hip_linwinds_perturbation_upload(icar_hip_ctx *c, int comp, const float *host)
{
    if (!c || !host) { icar_set_error("linwinds_perturbation_upload: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_linwinds_pert_copy(c, comp, const_cast<float *>(host), 1);
}

int icar_hip_spatial_winds(icar_hip_ctx *c, int update)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_spatial_winds_run(c, update);
}

size_t icar_hip_halo_count(const icar_hip_ctx *c, int dir, int halo)
{
    if (!c) return 0;
    if (dir == 0 || dir == 1) return (size_t)c->d.nx * c->d.nz * halo;
    return (size_t)halo * c->d.nz * c->d.ny;
}

int icar_hip_halo_pack(icar_hip_ctx *c, int dir, int halo, const int *fields, int nfields, void *dbuf)
{
    if (!c || !dbuf) { icar_set_error("halo_pack: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_halo_pack(c, dir, halo, fields, nfields, (float *)dbuf, false);
}

int icar_hip_halo_pack_dirs(icar_hip_ctx *c, int ndir, const int *dirs, int halo, const int *fields, int nfields, void *const *dbufs)
{
    if (!c || !dirs || !dbufs) { icar_set_error("halo_pack_dirs: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_halo_pack_dirs(c, ndir, dirs, halo, fields, nfields, dbufs, false);
}

int icar_hip_halo_unpack_dirs(icar_hip_ctx *c, int ndir, const int *dirs, int halo, const int *fields, int nfields, void *const *dbufs)
{
    if (!c || !dirs || !dbufs) { icar_set_error("halo_unpack_dirs: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_halo_pack_dirs(c, ndir, dirs, halo, fields, nfields, dbufs, true);
}

int icar_hip_halo_unpack(icar_hip_ctx *c, int dir, int halo, const int *fields, int nfields, const void *dbuf)
{
    if (!c || !dbuf) { icar_set_error("halo_unpack: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_halo_pack(c, dir, halo, fields, nfields, (float *)dbuf, true);
}

// ---- second stream -------------------------------------------------------------------------------------------------
// aux_fork : the aux stream waits for everything issued on the main stream so far
// aux_begin / aux_end : entry points called in between launch on the aux stream
// aux_join : the main stream waits for everything issued on the aux stream so far
static int ensure_aux(icar_hip_ctx *c)
{
    if (c->aux) return 0;
    // same priority as the main stream: the strips are issued first and start first; a lowest-priority aux stream measured
    // 1.5 % slower per step (the interior launch is the bulk of the step's work and must not be throttled)
    HIPCHK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    return 0;
}

int icar_hip_aux_fork(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (c->on_aux) { icar_set_error("aux_fork: already on the aux stream"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (ensure_aux(c)) return 1;
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    return 0;
}

int icar_hip_aux_begin(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (c->on_aux) { icar_set_error("aux_begin: already on the aux stream"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    if (ensure_aux(c)) return 1;
    c->main_saved = c->stream; c->stream = c->aux; c->on_aux = true;
    return 0;
}

int icar_hip_aux_end(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (!c->on_aux) { icar_set_error("aux_end: not on the aux stream"); return 1; }
    c->stream = c->main_saved; c->on_aux = false;
    return 0;
}

int icar_hip_aux_join(icar_hip_ctx *c)
{
    if (!c) { icar_set_error("null ctx"); return 1; }
    if (c->on_aux) { icar_set_error("aux_join: call aux_end first"); return 1; }
    if (!c->aux) return 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev_join, c->aux));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return 0;
}

int icar_hip_timing_enable(icar_hip_ctx *c, int on) { if (!c) return 1; hipSetDevice(c->device); c->timing = on != 0; return 0; }
int icar_hip_timing_groups(icar_hip_ctx *c, const char *csv)
{
    if (!c) return 1;
    c->timing_only = (csv && *csv) ? std::string(",") + csv + "," : std::string();
    return 0;
}
int icar_hip_timing_reset(icar_hip_ctx *c) { if (!c) return 1; hipSetDevice(c->device); hipStreamSynchronize(c->stream); drain_timers(c); c->timers.clear(); return 0; }
int icar_hip_timing_read(icar_hip_ctx *c, const char *group, double *total_ms, int *launches)
{
    if (!c || !group) return 1;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drain_timers(c);
    auto it = c->timers.find(group);
    if (total_ms) *total_ms = (it == c->timers.end()) ? 0.0 : it->second.total_ms;
    if (launches) *launches = (it == c->timers.end()) ? 0 : it->second.launches;
    return 0;
}

}  // extern "C"
