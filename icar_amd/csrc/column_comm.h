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
    __device__ __forceinline__ bool any(bool p) { return __any(p); }
    // min over levels >= own (inactive top lanes pass the neutral element)
    __device__ __forceinline__ double suffix_min(double v, int = 0)
    {
        for (int dd = 1; dd < 64; dd <<= 1) {
            const double o_ = __shfl_down(v, dd);
            if (k + dd < 64) v = fmin(v, o_);
        }
        return v;
    }
    // nearest level >= own that "has" the species (inactive top lanes have has=1, value 0 == vtXk(kte+1) = 0)
    __device__ __forceinline__ void carry_down2(float &a, float &b, int has)
    {
        for (int dd = 1; dd < 64; dd <<= 1) {
            const float oa = __shfl_down(a, dd), ob = __shfl_down(b, dd);
            const int oh = __shfl_down(has, dd);
            if (!has && k + dd < 64) { a = oa; b = ob; has = oh; }
        }
        if (!has) { a = 0.f; b = 0.f; }
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
    __device__ __forceinline__ int loop_max(int n) { return n; }           // one column per wave: already uniform
    __device__ __forceinline__ void up2(float x, float y, float &ux, float &uy) { ux = __shfl_down(x, 1); uy = __shfl_down(y, 1); }
    __device__ __forceinline__ float up1(float x) { return __shfl_down(x, 1); }
    // the same three services for several species at once (BlockComm pays one barrier round for all of them)
    __device__ __forceinline__ void carry_down2x2(float &a0, float &b0, int has0, float &a1, float &b1, int has1, int = 0)
    { carry_down2(a0, b0, has0); carry_down2(a1, b1, has1); }
    // nblk[s]: the longest sub-step loop of species s among the columns this communicator spans (here: the one column)
    __device__ __forceinline__ void sed_plan4(const int cond[4], const int ns[4], int kte, int ksed1[4], float onstep[4], int nblk[4])
    { for (int s = 0; s < 4; ++s) { sed_plan(cond[s], ns[s], kte, ksed1[s], onstep[s]); nblk[s] = (int)lroundf(1.f / onstep[s]); } }
    __device__ __forceinline__ void up6(const float v[6], float u[6]) { for (int s = 0; s < 6; ++s) u[s] = __shfl_down(v[s], 1); }
    __device__ __forceinline__ void up6_of(const float v[6], float u[6], unsigned slots)
    { for (int s = 0; s < 6; ++s) if (slots & (1u << s)) u[s] = __shfl_down(v[s], 1); }
};

// cpb whole columns per block of nt >= cpb*nz threads; thread = level*cpb + column.  LDS (dynamic):
// double d[nt] | float f[2][6][nt] | int has[nt] | int colmax[2][cpb+1] | int blkmax
struct BlockComm {
    double *sd; float *sf; int *shas, *scolmax, *sblkmax;
    int tid, nt, k, col, cpb, nz; bool active; unsigned step;
    __host__ __device__ static size_t lds_bytes(int nt, int cpb) { return (size_t)nt * (8 + 48 + 4) + (size_t)(2 * (cpb + 1) + 1) * 4; }
    // col_ok: this thread's column lies inside the tile (blocks are aligned to multiples of cpb columns)
    __device__ __forceinline__ BlockComm(void *lds, int tid_, int nt_, int cpb_, int nz_, int col_lo, int col_hi)
        : tid(tid_), nt(nt_), cpb(cpb_), nz(nz_), step(0)
    {
        sd = (double *)lds; sf = (float *)(sd + nt); shas = (int *)(sf + 12 * nt); scolmax = shas + nt; sblkmax = scolmax + 2 * (cpb + 1);
        const bool in = tid < cpb * nz;
        k = in ? tid / cpb : nz - 1;            // idle threads sit at "kte" of a dummy column: they never read upward
        col = in ? tid - k * cpb : cpb;
        active = in && col >= col_lo && col <= col_hi;
        if (!active) { k = nz - 1; }
    }
    __device__ __forceinline__ float &F(int b, int w, int t) { return sf[(b * 6 + w) * nt + t]; }
    __device__ __forceinline__ bool any(bool p) { return __syncthreads_or(p); }
    __device__ __forceinline__ double suffix_min(double v, int = 0)
    {
        // two-level: minimum over the rest of my chunk of 8 levels, then over the chunk minima above (<= 7 + nz/8 reads
        // instead of up to nz-1; a wave pays for its lowest lane).  min is exact, so the grouping does not matter.
        double *cm = (double *)sf;                                   // chunk minima [chunk][cpb], 6 nt doubles available
        __syncthreads();
        sd[tid] = v;
        __syncthreads();
        const int c8 = k >> 3, kend = min((c8 + 1) << 3, nz);
        if (active) for (int kk = k + 1; kk < kend; ++kk) v = fmin(v, sd[kk * cpb + col]);
        if (active && (k & 7) == 0) cm[c8 * cpb + col] = v;          // the lowest level of a chunk now holds its minimum
        __syncthreads();
        if (active) for (int cc = c8 + 1; cc * 8 < nz; ++cc) v = fmin(v, cm[cc * cpb + col]);
        step = 0;                                                    // the f area was used: next up*() starts fresh
        return v;
    }
    __device__ __forceinline__ void carry_down2(float &a, float &b, int has)
    {
        __syncthreads();
        F(0, 0, tid) = a; F(0, 1, tid) = b;
        if (nz <= 64) {
            // per-column bit mask of the levels that hold the species: the nearest one above is a shift + count-trailing-
            // zeros away (a scan loop runs, for the whole wave, as long as its unluckiest lane: up to nz iterations)
            unsigned long long *mask = (unsigned long long *)scolmax;             // (cpb+1) 64-bit words
            if (tid <= cpb) mask[tid] = 0ull;
            __syncthreads();
            if (has && active) atomicOr(&mask[col], 1ull << k);
            __syncthreads();
            if (!has) {                          // only active threads can have has == 0
                const unsigned long long m = (k + 1 < 64) ? (mask[col] >> (k + 1)) : 0ull;
                if (m) { const int kk = k + 1 + __builtin_ctzll(m); a = F(0, 0, kk * cpb + col); b = F(0, 1, kk * cpb + col); }
                else { a = 0.f; b = 0.f; }
            }
        } else {
            shas[tid] = has;
            __syncthreads();
            if (!has) {
                int kk = k + 1;
                while (kk < nz && !shas[kk * cpb + col]) ++kk;
                if (kk < nz) { a = F(0, 0, kk * cpb + col); b = F(0, 1, kk * cpb + col); }
                else { a = 0.f; b = 0.f; }
            }
        }
        step = 0;                                // f[0] was just used: the next up*() starts on f[1]
    }
    __device__ __forceinline__ void sed_plan(int cond, int ns, int kte, int &ksed1, float &onstep)
    {
        __syncthreads();
        if (tid <= cpb) { scolmax[tid] = 0; scolmax[cpb + 1 + tid] = 0; }
        __syncthreads();
        if (cond) atomicMax(&scolmax[col], k);
        if (ns > 0) atomicMax(&scolmax[cpb + 1 + col], ns);
        __syncthreads();
        ksed1 = scolmax[col]; ns = scolmax[cpb + 1 + col];
        if (ksed1 == kte) ksed1 = kte - 1;
        onstep = (ns > 0) ? 1.f / (float)ns : 1.0f;
    }
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
    // value(s) of the level above in my column; alternating buffers => one barrier per sub-step
    __device__ __forceinline__ void up2(float x, float y, float &ux, float &uy)
    {
        const int b = (int)((++step) & 1u);
        F(b, 0, tid) = x; F(b, 1, tid) = y;
        __syncthreads();
        const bool u = active && k + 1 < nz;
        ux = u ? F(b, 0, tid + cpb) : 0.f; uy = u ? F(b, 1, tid + cpb) : 0.f;
    }
    __device__ __forceinline__ float up1(float x)
    {
        const int b = (int)((++step) & 1u);
        F(b, 0, tid) = x;
        __syncthreads();
        return (active && k + 1 < nz) ? F(b, 0, tid + cpb) : 0.f;
    }
    // two species in one exchange (nz <= 64: per-column bit masks in the sd area; else two plain calls)
    __device__ __forceinline__ void carry_down2x2(float &a0, float &b0, int has0, float &a1, float &b1, int has1, int = 0)
    {
        if (nz > 64) { carry_down2(a0, b0, has0); carry_down2(a1, b1, has1); return; }
        __syncthreads();
        F(0, 0, tid) = a0; F(0, 1, tid) = b0; F(0, 2, tid) = a1; F(0, 3, tid) = b1;
        unsigned long long *mask = (unsigned long long *)sd;                      // 2 x (cpb+1) words, nt >= 2(cpb+1)
        if (tid < 2 * (cpb + 1)) mask[tid] = 0ull;
        __syncthreads();
        if (active) {
            if (has0) atomicOr(&mask[col], 1ull << k);
            if (has1) atomicOr(&mask[cpb + 1 + col], 1ull << k);
        }
        __syncthreads();
        if (!has0) {
            const unsigned long long m = (k + 1 < 64) ? (mask[col] >> (k + 1)) : 0ull;
            if (m) { const int kk = k + 1 + __builtin_ctzll(m); a0 = F(0, 0, kk * cpb + col); b0 = F(0, 1, kk * cpb + col); }
            else { a0 = 0.f; b0 = 0.f; }
        }
        if (!has1) {
            const unsigned long long m = (k + 1 < 64) ? (mask[cpb + 1 + col] >> (k + 1)) : 0ull;
            if (m) { const int kk = k + 1 + __builtin_ctzll(m); a1 = F(0, 2, kk * cpb + col); b1 = F(0, 3, kk * cpb + col); }
            else { a1 = 0.f; b1 = 0.f; }
        }
        step = 0;
    }
    // four sedimentation plans in one exchange: per-column max level with a sedimenting particle and max sub-step count
    // nblk[s]: the block's longest sub-step loop of species s = the maximum over its columns of what nstep will be (max(ns, 1)),
    // read from the per-column maxima that are in LDS anyway -- no reduction round of its own
    __device__ __forceinline__ void sed_plan4(const int cond[4], const int ns[4], int kte, int ksed1[4], float onstep[4], int nblk[4])
    {
        int *cm = shas;                                                           // 8 x (cpb+1) ints, nt >= 8(cpb+1) for nz >= 9
        if (8 * (cpb + 1) > nt) {
            for (int s = 0; s < 4; ++s) { sed_plan(cond[s], ns[s], kte, ksed1[s], onstep[s]); nblk[s] = loop_max((int)lroundf(1.f / onstep[s])); }
            return;
        }
        // (measured: these per-thread LDS atomics -- ~10 lanes of a wave per word -- beat both a gather by 8 cpb leader threads
        // looping over the levels, 1.88 -> 2.03 ms, and a pre-reduction with wave shifts by cpb, 2 cpb, ..., 2.05 ms; only atomics of a
        // WHOLE wave on one word are worth avoiding, see loop_max)
        __syncthreads();
        if (tid < 8 * (cpb + 1)) cm[tid] = 0;
        __syncthreads();
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
    __device__ __forceinline__ void up6(const float v[6], float u[6])
    {
        const int b = (int)((++step) & 1u);
        for (int s = 0; s < 6; ++s) F(b, s, tid) = v[s];
        __syncthreads();
        const bool up = active && k + 1 < nz;
        for (int s = 0; s < 6; ++s) u[s] = up ? F(b, s, tid + cpb) : 0.f;
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
