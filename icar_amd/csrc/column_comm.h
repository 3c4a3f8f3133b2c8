// icar_amd/csrc/column_comm.h -- how the levels of a column talk to each other when every thread owns ONE level.
//   WaveComm  : one column per wave, level = lane; wave shuffles / ballot.  nz of 64 lanes are busy.
//   BlockComm : cpb whole columns per block, thread = level*cpb + column (column fastest: a wave spans few levels of
//               neighbouring columns => coalesced rows and little divergence); the couplings go through LDS.
// Used by the Thompson column (thompson_lane.inc) and by mp_simple (mp_simple.hip).
#pragma once
#include <hip/hip_runtime.h>

struct WaveComm {
    int k; bool active;
    __device__ __forceinline__ WaveComm(int lane, int nz) : k(lane), active(lane < nz) {}
    // min over levels >= own (inactive top lanes pass the neutral element)
    __device__ __forceinline__ double suffix_min(double v, int = 0)
    {
        for (int dd = 1; dd < 64; dd <<= 1) {
            const double o_ = __shfl_down(v, dd);
            if (k + dd < 64) v = fmin(v, o_);
        }
        return v;
    }
    // ksed1 = highest level with a sedimenting particle (kts if none; kte -> kte-1), onstep = 1/max(nstep) (:2548-2555)
    __device__ __forceinline__ void sed_plan(int cond, int ns, int kte, int &ksed1, float &onstep)
    {
        const unsigned long long m = __ballot(cond);
        ksed1 = m ? (63 - __clzll((long long)m)) : 0;
        for (int dd = 32; dd > 0; dd >>= 1) { const int o = __shfl_xor(ns, dd); ns = ns > o ? ns : o; }
        if (ksed1 == kte) ksed1 = kte - 1;
        onstep = (ns > 0) ? 1.f / (float)ns : 1.0f;
    }
    // nblk[s]: the longest sub-step loop of species s among the columns this communicator spans (here: the one column)
    __device__ __forceinline__ void sed_plan4(const int cond[4], const int ns[4], int kte, int ksed1[4], float onstep[4], int nblk[4])
    { for (int s = 0; s < 4; ++s) { sed_plan(cond[s], ns[s], kte, ksed1[s], onstep[s]); nblk[s] = (int)lroundf(1.f / onstep[s]); } }
    __device__ __forceinline__ void up6_of(const float v[6], float u[6], unsigned slots)
    { for (int s = 0; s < 6; ++s) if (slots & (1u << s)) u[s] = __shfl_down(v[s], 1); }

    // ---- Thompson's merged exchanges (the same interface as BlockComm's, see there) ----
    unsigned long long lm[5]; float pv[6]; double wmin1;
    __device__ __forceinline__ void th_init(double) {}
    __device__ __forceinline__ bool any_min(bool flag, double v, double &vmin) { vmin = suffix_min(v); return __any(flag); }
    __device__ __forceinline__ void post_fall1(double n0, const float v[5], const int has[5])
    {
        wmin1 = suffix_min(n0);
        for (int s = 0; s < 5; ++s) { pv[s] = v[s]; lm[s] = __ballot(active && has[s]); }
    }
    __device__ __forceinline__ double min_fall1() { return wmin1; }
    __device__ __forceinline__ void post_fall2(float ag) { pv[5] = ag; }
    // nearest level above kk0 of my column that posted flag s (-1: none); flag s of level kk; value `slot` of level kk.
    // peek() is a wave shuffle: every lane of the wave has to call it (no divergent control flow around it)
    __device__ __forceinline__ int above(int s, int kk0) const
    {
        const unsigned long long t = (kk0 + 1 < 64) ? (lm[s] >> (kk0 + 1)) : 0ull;
        return t ? kk0 + 1 + __builtin_ctzll(t) : -1;
    }
    __device__ __forceinline__ bool bit(int s, int kk) const { return (lm[s] >> kk) & 1ull; }
    __device__ __forceinline__ float peek(int slot, int kk) const { return __shfl(pv[slot], kk); }
    __device__ __forceinline__ void plan4(const int cond[4], const int ns[4], int kte, int ksed1[4], float onstep[4], int nblk[4])
    { sed_plan4(cond, ns, kte, ksed1, onstep, nblk); }
};

// cpb whole columns per block of nt >= cpb*nz threads; thread = level*cpb + column.  LDS (dynamic):
// double d[nt] | float f[2][6][nt] | int has[nt] | int colmax[2][cpb+1] | int blkmax | (8-byte aligned) the areas of Thompson's
// merged exchanges, each written once per kernel: double wmin[2][nt/64][cpb+1] | u64 mask[5][cpb+1] | int plan[8][cpb+1]
struct BlockComm {
    double *sd; float *sf; int *shas, *scolmax, *sblkmax;
    double *swmin; unsigned long long *smask; int *splan;
    int tid, nt, k, col, cpb, nz; bool active; unsigned step;
    __host__ __device__ static size_t base_bytes(int nt, int cpb) { return (((size_t)nt * (8 + 48 + 4) + (size_t)(2 * (cpb + 1) + 1) * 4) + 7) & ~(size_t)7; }
    __host__ __device__ static size_t lds_bytes(int nt, int cpb)
    { return base_bytes(nt, cpb) + (size_t)(cpb + 1) * ((size_t)2 * (nt / 64) * 8 + 5 * 8 + 8 * 4); }
    // col_ok: this thread's column lies inside the tile (blocks are aligned to multiples of cpb columns)
    __device__ __forceinline__ BlockComm(void *lds, int tid_, int nt_, int cpb_, int nz_, int col_lo, int col_hi)
        : tid(tid_), nt(nt_), cpb(cpb_), nz(nz_), step(0)
    {
        sd = (double *)lds; sf = (float *)(sd + nt); shas = (int *)(sf + 12 * nt); scolmax = shas + nt; sblkmax = scolmax + 2 * (cpb + 1);
        swmin = (double *)((char *)lds + base_bytes(nt, cpb)); smask = (unsigned long long *)(swmin + 2 * (nt / 64) * (cpb + 1));
        splan = (int *)(smask + 5 * (cpb + 1));
        const bool in = tid < cpb * nz;
        k = in ? tid / cpb : nz - 1;            // idle threads sit at "kte" of a dummy column: they never read upward
        col = in ? tid - k * cpb : cpb;
        active = in && col >= col_lo && col <= col_hi;
        if (!active) { k = nz - 1; }
    }
    __device__ __forceinline__ float &F(int b, int w, int t) { return sf[(b * 6 + w) * nt + t]; }
    // per-column OR / maximum of a non-negative float (bit patterns of non-negative floats order like integers)
    __device__ __forceinline__ bool col_any(bool p)
    {
        __syncthreads();
        if (tid <= cpb) scolmax[tid] = 0;
        __syncthreads();
        if (p) scolmax[col] = 1;
        __syncthreads();
        return scolmax[col] != 0;
    }
    __device__ __forceinline__ float col_max_pos(float v)
    {
        __syncthreads();
        if (tid <= cpb) scolmax[tid] = 0;
        __syncthreads();
        if (active) atomicMax(&scolmax[col], __float_as_int(v));
        __syncthreads();
        return __int_as_float(scolmax[col]);
    }
    // maximum over a wave first (butterfly): 64 lanes hitting ONE LDS address with an atomic serialise, ~15 cycles each -- a
    // block-wide atomicMax per thread cost a 240-thread block ~3.6 k cycles per call
    static __device__ __forceinline__ int wave_max(int v)
    {
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        return v;
    }
    __device__ __forceinline__ int loop_max(int n)
    {
        __syncthreads();
        if (tid == 0) *sblkmax = 0;
        __syncthreads();
        const int w = wave_max(active ? n : 0);
        if ((tid & 63) == 0) atomicMax(sblkmax, w);
        __syncthreads();
        return *sblkmax;
    }
    __device__ __forceinline__ float up1(float x)
    {
        const int b = (int)((++step) & 1u);
        F(b, 0, tid) = x;
        __syncthreads();
        return (active && k + 1 < nz) ? F(b, 0, tid + cpb) : 0.f;
    }
    // up6 restricted to the slots named in `slots` (block-uniform): the others are neither written nor read
    __device__ __forceinline__ void up6_of(const float v[6], float u[6], unsigned slots)
    {
        const int b = (int)((++step) & 1u);
        for (int s = 0; s < 6; ++s) if (slots & (1u << s)) F(b, s, tid) = v[s];
        __syncthreads();
        const bool up = active && k + 1 < nz;
        for (int s = 0; s < 6; ++s) if (slots & (1u << s)) u[s] = up ? F(b, s, tid + cpb) : 0.f;
    }

    // ---- Thompson's merged exchanges: ONE block barrier each (round 4: any 1 + suffix-min 3 + 3, carry-down 3 + 3, plan 3).
    // Every area is written once per kernel, so nothing has to be protected from an earlier use; the areas that are combined
    // with atomics are cleared by th_init() at the top of the kernel (the caller's first barrier follows it).
    __device__ __forceinline__ void th_init(double neutral)
    {
        const int nw = nt >> 6;
        for (int t = tid; t < 2 * nw * (cpb + 1); t += nt) swmin[t] = neutral;
        for (int t = tid; t < 5 * (cpb + 1); t += nt) smask[t] = 0ull;
        for (int t = tid; t < 8 * (cpb + 1); t += nt) splan[t] = 0;
    }
    // min over the levels >= own that sit in my WAVE (lanes l, l + cpb, l + 2 cpb ... are the higher levels of a lane's column);
    // the lowest level a wave holds of a column (lanes < cpb) leaves the wave's minimum for the waves below
    __device__ __forceinline__ double wave_suffix_min(double v, int which)
    {
        const int lane = tid & 63;
        for (int d = cpb; d < 64; d <<= 1) { const double o = __shfl_down(v, d); if (lane + d < 64) v = fmin(v, o); }
        if (lane < cpb) swmin[(which * (nt >> 6) + (tid >> 6)) * (cpb + 1) + tid % cpb] = v;
        return v;
    }
    __device__ __forceinline__ double cross_wave_min(double v, int which) const
    {
        const int nw = nt >> 6, c = tid % cpb;
        for (int w = (tid >> 6) + 1; w < nw; ++w) v = fmin(v, swmin[(which * nw + w) * (cpb + 1) + c]);
        return v;
    }
    // block-wide OR of `flag` and the suffix-min of v (inactive threads pass the neutral element) behind the same barrier
    __device__ __forceinline__ bool any_min(bool flag, double v, double &vmin)
    {
        v = wave_suffix_min(v, 0);
        const bool any = __syncthreads_or(flag);
        vmin = any ? cross_wave_min(v, 0) : v;
        return any;
    }
    // first fall-speed exchange: the second graupel chain's suffix-min, the five carried values (rain mass / number, ice mass /
    // number, snow before its max with the rain speed) and five flags per level (has rain / ice / snow / graupel, T > T_0)
    double wmin1;
    __device__ __forceinline__ void post_fall1(double n0, const float v[5], const int has[5])
    {
        wmin1 = wave_suffix_min(n0, 1);
        for (int s = 0; s < 5; ++s) F(0, s, tid) = v[s];
        if (nz <= 64) {
            if (active) for (int s = 0; s < 5; ++s) if (has[s]) atomicOr(&smask[s * (cpb + 1) + col], 1ull << k);
        } else {
            int f = 0;
            for (int s = 0; s < 5; ++s) f |= (active && has[s]) ? (1 << s) : 0;
            shas[tid] = f;
        }
        __syncthreads();
    }
    __device__ __forceinline__ double min_fall1() const { return cross_wave_min(wmin1, 1); }
    __device__ __forceinline__ void post_fall2(float ag) { F(1, 0, tid) = ag; __syncthreads(); }
    __device__ __forceinline__ int above(int s, int kk0) const
    {
        if (nz <= 64) {
            const unsigned long long t = (kk0 + 1 < 64) ? (smask[s * (cpb + 1) + col] >> (kk0 + 1)) : 0ull;
            return t ? kk0 + 1 + __builtin_ctzll(t) : -1;
        }
        if (col >= cpb) return -1;
        int kk = kk0 + 1;
        while (kk < nz && !(shas[kk * cpb + col] & (1 << s))) ++kk;
        return kk < nz ? kk : -1;
    }
    __device__ __forceinline__ bool bit(int s, int kk) const
    {
        if (nz <= 64) return (smask[s * (cpb + 1) + col] >> kk) & 1ull;
        return col < cpb && (shas[kk * cpb + col] & (1 << s));
    }
    // value `slot` (0..4: post_fall1, 5: post_fall2) of level kk of my column
    __device__ __forceinline__ float peek(int slot, int kk) const
    { const int t = min(kk * cpb + col, nt - 1); return slot < 5 ? sf[slot * nt + t] : sf[6 * nt + t]; }
    // the four sedimentation plans (see sed_plan4) with the maxima collected in the area th_init() cleared: one barrier
    __device__ __forceinline__ void plan4(const int cond[4], const int ns[4], int kte, int ksed1[4], float onstep[4], int nblk[4])
    {
        int *cm = splan;
        for (int s = 0; s < 4; ++s) {
            if (cond[s]) atomicMax(&cm[(2 * s) * (cpb + 1) + col], k);
            if (ns[s] > 0) atomicMax(&cm[(2 * s + 1) * (cpb + 1) + col], ns[s]);
        }
        __syncthreads();
        for (int s = 0; s < 4; ++s) {
            int ks = cm[(2 * s) * (cpb + 1) + col]; const int n = cm[(2 * s + 1) * (cpb + 1) + col];
            if (ks == kte) ks = kte - 1;
            ksed1[s] = ks; onstep[s] = (n > 0) ? 1.f / (float)n : 1.0f;
            int nb = 1;
            for (int c = 0; c < cpb; ++c) nb = max(nb, cm[(2 * s + 1) * (cpb + 1) + c]);      // (columns outside the tile hold 0)
            nblk[s] = nb;
        }
    }
};


// Block geometry for BlockComm: cpb = floor(nt/nz) whole columns per nt-thread block.  nt is a multiple of 256 (4 waves
// per SIMD step): 320- or 640-thread blocks (5 / 10 waves) load the CU's SIMDs unevenly and measured 1.5x slower than
// 256.  A bigger block is taken only for a clear (>10 %) gain in busy threads, because its barriers cost more.
// Returns the fraction of busy threads (0 when nz > 1024).
static inline float block_comm_geometry(int nz, int &nt, int &cpb)
{
    // the smallest block wins unless a bigger one fills clearly better (measured at nz = 40, Thompson after round 2's work removal:
    // 6 columns per 256-thread block 1.98 ms, 12 per 512-thread block 2.08 ms, 25 per 1024-thread block 2.50 ms -- round 1's
    // kernel had 512 ahead by 2.5 %)
    float best = 0.0f; nt = 0; cpb = 0;
    for (int t = 256; t <= 1024; t *= 2) {
        if (t < nz) continue;
        const float u = (float)((t / nz) * nz) / t;
        const float need = !nt ? 0.0f : 0.10f;
        if (u > best + need) { best = u; nt = t; cpb = t / nz; }
    }
    return best;
}
