// icar_amd/csrc/mpdata.hip -- MPDATA (order >= 2) with flux-corrected transport as ONE kernel per corrective iteration
// (rows A2-A5 of SURVEY.md section 8: adv_mpdata.f90:44-105 donor cell, :107-255 pseudo-velocities, :383-385 scaling,
// adv_mpdata_FCT_core.f90:47-116 limiter, :389 final donor-cell pass).
//
// Round 1 ran these as three kernels that exchanged q2, u2, v2, w2 through HBM (7.4 GB per call against 0.92 GB of
// algorithmic traffic) and evaluated every limited face from scratch (700 VALU instructions per scalar-cell, 22 IEEE
// divisions).  This kernel reads each scalar once and writes it once:
//
//   * a block owns an x-tile of 64 lanes (58 outputs + 3 halo lanes per side) over ALL levels and MARCHES along y.
//     A thread owns KB consecutive levels of one column-of-the-plane; a wave is one group of KB levels.
//   * every intermediate of the scheme lives in registers as a rolling window over the planes:
//         step P:  q(P+2) arrives (requested right after the donor-cell pass of step P-1: the only read of the scalar)
//                  q2(P+1)        = donor-cell pass        (needs q at P, P+1, P+2)
//                  v2(P+1/2), Fy  = pseudo-velocity + unlimited flux of the y face between planes P and P+1
//                  u2(P), w2(P), Fx, Fz, beta_x(P), beta_z(P), limited x/z fluxes of plane P
//                  beta_y(P), limited y flux (P-1/2)
//                  out(P-1) is stored                      (the only write)
//   * x neighbours come from the neighbouring lanes by DPP wave shifts (v_*_dpp wave_shr:1 / wave_shl:1: one VALU
//     slot, mostly folded into the consuming instruction; ds_bpermute costs 24 cycles per wave on gfx950),
//     z neighbours inside a thread's own levels are registers, across threads they go through LDS twice per plane
//     (the pass-1 field of the two edge levels, then the limiter's beta of the two edge levels), y neighbours are the
//     rolling registers.
//   * the limiter is evaluated per CELL and direction (beta_in, beta_out: adv_mpdata_FCT_core.f90's carried variables
//     written as what they are -- properties of a cell), so every flux, extremum and quotient is computed once; the
//     limited flux of a face is min(1, beta, beta) times its unlimited flux (flux1 is linear in the velocity).
//   * quotients are n * v_rcp_f32(d) (1 ulp): the scheme's 22 divisions per scalar-cell cost 11 instead of 50 cycles
//     per wave each (profiles/micro/valubench.hip).  flux1(l, r, U) = ((U+|U|) l + (U-|U|) r)/2 is evaluated as
//     U * (U > 0 ? l : r), which is the same number.  Results agree with the CPU reference to ~1e-6 of the local field
//     scale; tests assert 1e-5 on every cell (north-star tolerance) -- the donor-cell kernel of the upwind scheme
//     (advect.hip) stays bit-exact.
//
// Everything that does not depend on the scalar is computed ONCE per step by k_mpdata_coef (icar_hip_setup_winds) and loaded:
// per face the antidiffusive coefficient |U|(1-|U|/Gbar)/2 and the two cross-term factors U Ubar_perp / (8 Gbar) (the six 4-point
// transverse Courant averages, the 1/(G_i + G_i-1), the ground / top / x-ring zeros folded in, the z faces already times dz), and
// per cell 1/(jaco rho), 1/(jaco rho dz) -- three float4 arrays (round 2 recomputed all of it for each of the 9 scalars:
// ~50 of 252 VALU instructions per scalar-cell).  They and U_m, V_m, W_m are re-read per scalar from L2 (64 B per cell against
// the 8 B of the scalar itself), which is why blocks of the same (tile, chunk) and different scalars are scheduled onto the
// same XCD.
#include "ctx.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

#define MP_HL 3                       // halo lanes per side
#define MP_XOUT (64 - 2 * MP_HL)      // outputs per 64-lane tile
#define MP_NW 8                       // waves per block at most: two per SIMD, ~250 VGPRs each
#define MP_KB 5                       // levels per thread at most
#define MP_ZH 2                       // halo levels of a level range: a fake edge corrupts the outputs of the 2 levels next to it
#define EPSQ 1e-10f
#define EPSF 1e-15f

// bound_ctrl:1 -- a lane without a source reads 0, so no `old` value has to be materialised and the shift can fold into
// the consuming VALU instruction (v_sub_f32_dpp ...)
__device__ __forceinline__ float dpp_l(float x)   // value of lane-1 (0 in lane 0)
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_r(float x)   // value of lane+1 (0 in lane 63)
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float upw(float l, float r, float U) { return U * (U > 0.0f ? l : r); }      // == flux1 (adv_mpdata.f90:40)
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
// Global accesses are raw buffer loads / stores: descriptor (4 SGPRs per array) + per-lane byte offset (one VGPR, the
// column) + scalar byte offset (plane and level, wave-uniform).  With flat addressing every one of the ~80 loads of a step
// paid a 64-bit VALU add for its address (v_lshl_add_u64).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t mkrsrc(const float *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, -1, 0x00020000); }
__device__ __forceinline__ float ldb(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void stb(rsrc_t r, int voff, int soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0); }
// ------------------------------------------------------------------------------------------------
// scalar-independent coefficients of the corrective iteration (mpdata_fluxes, adv_mpdata.f90:107-255), once per step
// eleven arrays of the tile's shape in one buffer (MPC_* = index of the array):
//   x face (i-1/2) of cell (i,k,j):        au, cuv, cuw
//   y face between j-1 and j:              av, cvu, cvw
//   z face above level k:                  (aw, cwu, cwv) * dz(k) ; zero for the top level (w2(kme) = 0, :214)
//   cell:                                  1 / (jaco rho), 1 / (jaco rho dz)
// a? = |C| (1 - 2 |C| / (G + G')) / 2 ;  c?? = C (sum of the 4 transverse Courant numbers around the face) / (16 (G + G'))
// with G = jaco [rho]; cross terms through the ground / column top (k-1, k+1 missing) and in the x ring are zero.
// ------------------------------------------------------------------------------------------------
enum { MPC_AU = 0, MPC_CUV, MPC_CUW, MPC_AV, MPC_CVU, MPC_CVW, MPC_AW, MPC_CWU, MPC_CWV, MPC_RDH, MPC_RDV, MPC_N };
template <bool RHO>
__global__ void __launch_bounds__(256)
k_mpdata_coef(Dims d, const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ Wz, const float *__restrict__ rho,
              const float *__restrict__ jaco, const float *__restrict__ dz, float *__restrict__ C)
{
    const size_t n3 = (size_t)d.nx * d.nz * d.ny;
    const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y, j = blockIdx.z;
    if (i >= d.nx || k >= d.nz) return;
    const int nx = d.nx, nz = d.nz, ny = d.ny;
    const int iL = max(i - 1, 0), iR = min(i + 1, nx - 1), jP = max(j - 1, 0), jN = min(j + 1, ny - 1), kB = max(k - 1, 0), kT = min(k + 1, nz - 1);
    auto at = [&](int ii, int kk, int jj) { return ii + nx * (kk + nz * jj); };
    auto G = [&](int ii, int kk, int jj) { const int c = at(ii, kk, jj); return RHO ? jaco[c] * rho[c] : jaco[c]; };
    const int c = at(i, k, j);
    const bool xin = (i > 0) && (i < nx - 1), kin = (k > 0) && (k < nz - 1);
    const float g = G(i, k, j), dzc = dz[c];
    // x face (i-1/2)
    {
        const float rG = frcp(g + G(iL, k, j));
        const float evv = (V[c] + V[at(i, k, jN)]) + (V[at(iL, k, j)] + V[at(iL, k, jN)]);
        const float evw = (Wz[c] + Wz[at(i, kB, j)]) + (Wz[at(iL, k, j)] + Wz[at(iL, kB, j)]);
        const float Uc = U[c], aU = fabsf(Uc), c0 = 0.0625f * Uc * rG;
        C[MPC_AU * n3 + c] = 0.5f * aU * (1.0f - 2.0f * aU * rG); C[MPC_CUV * n3 + c] = c0 * evv; C[MPC_CUW * n3 + c] = kin ? c0 * evw : 0.0f;
    }
    // y face between j-1 and j
    {
        const float rG = frcp(g + G(i, k, jP));
        const float evu = (U[at(i, k, jP)] + U[c]) + (U[at(iR, k, jP)] + U[at(iR, k, j)]);
        const float evw = (Wz[at(i, k, jP)] + Wz[at(i, kB, jP)]) + (Wz[c] + Wz[at(i, kB, j)]);
        const float Vc = V[c], aV = fabsf(Vc), c0 = 0.0625f * Vc * rG;
        C[MPC_AV * n3 + c] = 0.5f * aV * (1.0f - 2.0f * aV * rG); C[MPC_CVU * n3 + c] = xin ? c0 * evu : 0.0f; C[MPC_CVW * n3 + c] = kin ? c0 * evw : 0.0f;
    }
    // z face above level k
    if (k < nz - 1) {
        const float rG = frcp(g + G(i, kT, j));
        const float evu = (U[c] + U[at(i, kT, j)]) + (U[at(iR, k, j)] + U[at(iR, kT, j)]);
        const float evv = (V[c] + V[at(i, k, jN)]) + (V[at(i, kT, j)] + V[at(i, kT, jN)]);
        const float Wc = Wz[c], aW = fabsf(Wc), c0 = 0.0625f * Wc * rG;
        C[MPC_AW * n3 + c] = 0.5f * aW * (1.0f - 2.0f * aW * rG) * dzc; C[MPC_CWU * n3 + c] = xin ? c0 * evu * dzc : 0.0f; C[MPC_CWV * n3 + c] = c0 * evv * dzc;
    } else { C[MPC_AW * n3 + c] = 0.f; C[MPC_CWU * n3 + c] = 0.f; C[MPC_CWV * n3 + c] = 0.f; }
    C[MPC_RDH * n3 + c] = frcp(g); C[MPC_RDV * n3 + c] = frcp(dzc * g);
}

// KB levels per thread (at most MP_NW waves per block), FCT: limiter on (advect_density lives in the coefficients),
// PASS1: first corrective iteration (donor-cell pass inside); false: iord >= 3, where q2 == q (adv_mpdata.f90:393-402)
template <int KB, bool FCT, bool PASS1>
// 248 VGPRs, not the 256 two waves per SIMD could have (the attribute counts half of gfx90a+'s unified file: 124 -> 248).  The
// 248 blocks of a launch hold their CUs for the whole kernel, so whatever the host issues on the second stream beside the
// advection (whole-field forcing, the CFL reduction of the next update_dt) can only run in what these waves leave: with 2 x 248
// of a SIMD's 512 registers taken, one more wave of <= 16 VGPRs fits and those streaming kernels run concurrently; at 254 (what
// the allocator takes if allowed) they wait for the launch to end -- the advection alone is then 5 % faster, the step 2 % slower.
__global__ void __launch_bounds__(64 * MP_NW) __attribute__((amdgpu_num_vgpr(124)))
k_mpdata_fused(Dims d, CVarPtrs qin, VarPtrs qout,
               const float *__restrict__ Ug, const float *__restrict__ Vg, const float *__restrict__ Wg,
               const float *__restrict__ Cg, int asz, int clen, int ntile, int nchunk, int nscal, int nkr, int kstore)
{
    constexpr int H = KB + 2;                       // own levels + one halo level below and above
    // exchange slots, double-buffered by step parity (a step without plane-P work has only the first exchange)
    __shared__ float s_q2[2][MP_NW][2][64];         // pass-1 field of a wave's lowest / highest level
    __shared__ float s_bz[2][MP_NW][4][64];         // beta_in, beta_out of a wave's lowest / highest level
    // The part of the rolling window that belongs to plane M (= P-1) is produced at the end of a step and consumed in the
    // second half of the next one.  With 8 waves per block a thread may hold ~250 VGPRs, and the loads + arithmetic of the
    // first half of a step need that room to overlap: those 9 KB + 2 values per thread are parked in LDS in between
    // (thread-private float4 slots: no synchronisation, conflict-free b128 accesses).
    // The two z exchanges of a step are synchronised between NEIGHBOURING waves only: a wave needs the edge levels of the wave
    // below and the wave above it, nothing else.  A wave posts its edges (data, then -- lgkmcnt(0) in between -- a step counter in
    // s_flag), does the work that needs no neighbour, and spins on the two neighbouring counters.  With __syncthreads every step
    // ran at the pace of the slowest of the block's 8 waves, twice; now a late wave delays its two neighbours only, and they
    // catch up (1.30 -> 1.22 ms per advect() at 512 x 512 x 40, 9 scalars).  The double buffering by step parity stays sufficient: wave w overwrites a buffer at step
    // t + 2 only after it has seen the counters of w-1 / w+1 at t + 1, which they post after their reads of step t.
    __shared__ int s_flag[2][MP_NW];
    int seqA = 0, seqB = 0;
#define MP_POST(e, seq)                                                                                                   \
    {                                                                                                                      \
        ++seq;                                                                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                             \
        if (lane == 0) __hip_atomic_store(&s_flag[e][wv], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);            \
    }
#define MP_WAIT1(e, seq, w)                                                                                                \
    {                                                                                                                      \
        int spin = 0;                                                                                                      \
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_flag[e][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < seq) { \
            __builtin_amdgcn_s_sleep(1);                  /* 0, 1, 3: the same time */                                                                            \
            if (++spin > (1 << 26)) __builtin_trap();              /* a lost neighbour must not hang the device */          \
        }                                                                                                                  \
    }
#define MP_WAIT(e, seq)                                                                                                    \
    {                                                                                                                      \
        if (wv > 0) MP_WAIT1(e, seq, wv - 1)                                                                               \
        if (wv < nw - 1) MP_WAIT1(e, seq, wv + 1)                                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                                             \
    }
    constexpr bool PARK = true;
    constexpr int NA4 = (H + 3) / 4, NB4 = 2 * KB;  // float4 slots: q2M[H] | mM nM v2S FyS bYinM bYoutM acc rdhM [KB each]
    constexpr int NC4 = (2 * KB + 3) / 4;           // 1 / (jaco rho), 1 / (jaco rho dz) of plane N: loaded for the donor-cell pass, needed
                                                    // again by the roll of the NEXT step (N has become P by then)
    __shared__ float4 s_park[PARK ? NA4 + NB4 + NC4 : 1][PARK ? 64 * MP_NW : 1];

    const int lane = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y), nw = blockDim.y;
    const int tid = lane + 64 * wv;
    const int nx = d.nx, nz = d.nz, ny = d.ny;
    const int sj4 = 4 * d.sj;                       // bytes per plane (the host checks that a field is < 2 GiB)
    // work item.  Items are ordered group-major (group = (tile, chunk), then the scalars of that group) and dealt to the 8
    // XCDs in contiguous runs of `cap` items; consecutive block ids go to the XCDs round-robin, so block id = xcd + 8 * slot.
    // The scalars of a group therefore share an XCD (at most two groups per XCD are split) and start together: the
    // scalar-independent arrays come from HBM once per group and from that XCD's L2 for the other scalars.
    int tile, chunk, m, kr;
    {
        const unsigned id = blockIdx.x, nv = (unsigned)nscal, nitem = (unsigned)(ntile * nchunk * nkr) * nv;
        const unsigned cap = (nitem + 7u) / 8u, xcd = id & 7u, slot = id >> 3;
        const unsigned it = xcd * cap + slot;
        if (slot >= cap || it >= nitem) return;
        if (lane == 0) { s_flag[0][threadIdx.y] = 0; s_flag[1][threadIdx.y] = 0; }
        __syncthreads();
        const unsigned g = it / nv;
        m = (int)(it - g * nv);
        tile = (int)(g % (unsigned)ntile); chunk = (int)((g / (unsigned)ntile) % (unsigned)nchunk); kr = (int)(g / (unsigned)(ntile * nchunk));
    }
    // More levels than 8 waves x 5 can hold are split into level ranges, each its own work item: a range stores `kstore`
    // levels [ka, kb] and computes MP_ZH more on either side (whatever a fake edge corrupts stays inside those halo levels);
    // the flags of the real bottom / top are those of the global level index, so a range edge is not a boundary.
    const int ka = kr * kstore, kb = min(ka + kstore - 1, nz - 1), kbase = max(ka - MP_ZH, 0);
    const float *__restrict__ qp = qin.p[0];
    float *__restrict__ outp = qout.p[0];
#pragma unroll
    for (int mm = 1; mm < ICAR_MAX_ADV; ++mm) if (mm == m) { qp = qin.p[mm]; outp = qout.p[mm]; }   // (a dynamic index would put the tables in scratch)
    const rsrc_t q = mkrsrc(qp), out = mkrsrc(outp), Ur = mkrsrc(Ug), Vr = mkrsrc(Vg), Wr = mkrsrc(Wg),
                 cr = mkrsrc(Cg);                           // the eleven coefficient arrays, asz bytes each

    const int i = 1 - MP_HL + tile * MP_XOUT + lane;
    const int ic = min(max(i, 0), nx - 1);
    const bool xlo = (i == 0), xhi = (i == nx - 1), xin = (i > 0) && (i < nx - 1);
    const bool xring = xlo || xhi;
    const bool lane_out = (lane >= MP_HL) && (lane < 64 - MP_HL) && xin;
    const int bx = 4 * ic;
    const int ja = 1 + chunk * clen, jb = min(ja + clen - 1, ny - 2);
    const int k0 = kbase + wv * KB;
    // level of slot h (0..H-1) = k0-1+h, clamped for addressing; flags are wave-uniform
    int kc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) kc[h] = min(max(k0 - 1 + h, 0), nz - 1) * nx * 4;       // byte offset of the level

#define LDP(arr, h, plane) ldb(arr, bx, (plane) * sj4 + kc[h])
#define LDC(a, h, plane) ldb(cr, bx, (a) * asz + (plane) * sj4 + kc[h])                 /* coefficient array a (MPC_*) */
#define CLAMPJ(p) min(max((p), 0), ny - 1)

    // ---- rolling state (planes relative to the step's in-plane index P; N = P+1, M = P-1) ----
    float qP[KB], qN[H], qPh0 = 0.f, qPh1 = 0.f;            // q (= l of the limiter): plane P own levels (+ its halo levels), plane N
    float q2P[H];                                           // field after pass 1, plane P
    float mP[KB], nP[KB];                                   // max / min of (q2, l) per cell, plane P
    float DxP[KB], SxP[KB], DzP[KB], SzP[KB];               // q2(i+1) -+ q2(i-1), q2(k+1) -+ q2(k-1) on plane P
    // plane-M part (parked between steps): pkA = q2M[H] ; pkB = mM nM v2S FyS bYinM bYoutM acc rdhM, KB values each
    //   mM, nM: extrema of plane M; v2S, FyS: pseudo-velocity / unlimited flux of the y face (P-1/2); bY*M: beta_y of
    //   plane M; acc: q2 - x/z/south contributions of plane M; rdhM: 1 / (jaco rho) of plane M
    float pkA[NA4 * 4], pkB[NB4 * 4];
#pragma unroll
    for (int t = 0; t < NA4 * 4; ++t) pkA[t] = 0.f;
#pragma unroll
    for (int t = 0; t < NB4 * 4; ++t) pkB[t] = 0.f;
    if (PARK) {
#pragma unroll
        for (int t = 0; t < NA4 + NB4 + NC4; ++t) s_park[t][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) { mP[kk] = nP[kk] = 0.f; DxP[kk] = SxP[kk] = DzP[kk] = SzP[kk] = 0.f; }
#pragma unroll
    for (int h = 0; h < H; ++h) q2P[h] = 0.f;
    const int P0 = ja - 3;
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) qP[kk] = LDP(q, kk + 1, CLAMPJ(P0));
#pragma unroll
    for (int h = 0; h < H; ++h) qN[h] = LDP(q, h, CLAMPJ(P0 + 1));

    // inputs of one step (plane indices relative to that step's P): see ISSUE_LOADS
    float qNN[H], WN[KB + 1], UN[KB], VN[KB], VNN[KB], rdhN[KB], rdvN[KB];
    float avN[KB], cvuN[KB], cvwN[KB], auP[KB], cuvP[KB], cuwP[KB], awP[KB + 1], cwuP[KB + 1], cwvP[KB + 1];
// group A: what the donor-cell pass (first half of a step) reads; group B: the x / y face coefficients (plane P / face P | N),
// requested at the top of the step; group Z: the z face coefficients of plane P, requested after the donor-cell pass
#define ISSUE_LOADS_A(PP)                                                                                                \
    {                                                                                                                    \
        const int lN = CLAMPJ((PP) + 1), lNN = CLAMPJ((PP) + 2);                                                         \
        _Pragma("unroll") for (int h = 0; h <= KB; ++h) WN[h] = LDP(Wr, h, lN);                                          \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { UN[kk] = LDP(Ur, kk + 1, lN); VN[kk] = LDP(Vr, kk + 1, lN); VNN[kk] = LDP(Vr, kk + 1, lNN); } \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { rdhN[kk] = LDC(MPC_RDH, kk + 1, lN); rdvN[kk] = LDC(MPC_RDV, kk + 1, lN); } \
    }
#define ISSUE_LOADS_B(PP)                                                                                                \
    {                                                                                                                    \
        const int lP = CLAMPJ(PP), lN = CLAMPJ((PP) + 1);                                                                \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { avN[kk] = LDC(MPC_AV, kk + 1, lN); cvuN[kk] = LDC(MPC_CVU, kk + 1, lN); cvwN[kk] = LDC(MPC_CVW, kk + 1, lN); } \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { auP[kk] = LDC(MPC_AU, kk + 1, lP); cuvP[kk] = LDC(MPC_CUV, kk + 1, lP); cuwP[kk] = LDC(MPC_CUW, kk + 1, lP); } \
    }
#define ISSUE_LOADS_Z(PP)                                                                                                \
    {                                                                                                                    \
        const int lP = CLAMPJ(PP);                                                                                       \
        _Pragma("unroll") for (int h = 0; h <= KB; ++h) { awP[h] = LDC(MPC_AW, h, lP); cwuP[h] = LDC(MPC_CWU, h, lP); cwvP[h] = LDC(MPC_CWV, h, lP); } \
    }
// the scalar itself: every plane of it comes from HBM exactly once, so its latency is the longest of all inputs
#define ISSUE_LOADS_Q(DST, PLANE)                                                                                        \
    {                                                                                                                    \
        const int lq = CLAMPJ(PLANE);                                                                                    \
        _Pragma("unroll") for (int h = 0; h < H; ++h) DST[h] = LDP(q, h, lq);                                            \
    }
    ISSUE_LOADS_Q(qNN, P0 + 2)
    ISSUE_LOADS_A(P0)
    // One step of the march.  STEADY = every stage is on and no plane of the window is a boundary row of the domain
    // (the bulk of a chunk): the stage conditions and the first/last-row forms of the y limiter are compiled out, and
    // with them the zero-initialisations and merge copies of ~70 values per step.  The generic form runs the warm-up
    // steps of a chunk, its last steps and the chunks that touch row 0 / ny-1.
    // Level flags.  "Is slot h the ground / the top of the column" is wave-uniform and loop-invariant; written as k-comparisons
    // per slot the compiler keeps every one of them as a 64-bit select mask in SGPRs (~50 pairs, half of them spilled to
    // VGPR lanes and read back with v_readlane every step).  Most of them are not needed at all:
    //   * the halo slots are CLAMPED loads, so the slot above the top level holds the top level's own values and the slot
    //     below level 0 those of level 0: upw(q(top), q(top+1), W) IS q(top) W (adv_mpdata.f90:96), and the z limiter's
    //     first-cell form (extrema without the cell below, no flux through the ground) falls out of the general one because
    //     max(q2(0), l(0)) is the cell's own extremum and the ground flux is +-0;
    //   * what is left -- zero pseudo-velocity through the ground and the top face, no z cross terms in the lowest and highest
    //     level, the last-cell form of the z limiter, the store range -- is decided from four scalars per step (ground, slot of the
    //     top level, first / last stored slot), made opaque below so that the comparisons are redone by the scalar unit inside the
    //     step instead of living in SGPR pairs across it.
    const int htop_u = __builtin_amdgcn_readfirstlane(nz - k0), gnd_u = __builtin_amdgcn_readfirstlane((k0 == 0) ? 1 : 0),    // slot h = level k0-1+h
              kst0_u = __builtin_amdgcn_readfirstlane(ka - k0), kst1_u = __builtin_amdgcn_readfirstlane(kb - k0);
    auto step = [&](auto steady_tag, const int P) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        int htop = htop_u, gnd = gnd_u, kst0 = kst0_u, kst1 = kst1_u;
        asm volatile("" : "+s"(htop), "+s"(gnd), "+s"(kst0), "+s"(kst1));
        const int N = P + 1;
        const int pP = CLAMPJ(P), pN = CLAMPJ(N), pNN = CLAMPJ(N + 1);
        const int par = (P - P0) & 1;
        const bool haveN = STEADY || ((N >= 0) && (N <= ny - 1));
        // Group A of this step's global loads was issued during the previous step (after its x/z limiter), group B is
        // issued here; each group back to back -- loads placed next to their use were waited for one by one: 59 exposed
        // L2 round trips per step.
        ISSUE_LOADS_B(P)                                           // land while the donor-cell pass runs
        __builtin_amdgcn_sched_barrier(0);

        // ================= S1: donor-cell pass on plane N, its extrema and x/z differences =================
        float q2N[H], mN[KB], nN[KB], DxN[KB], SxN[KB], DzN[KB], SzN[KB];
#pragma unroll
        for (int h = 0; h < H; ++h) q2N[h] = qN[h];
        if (haveN) {
            const bool ring = !STEADY && ((N == 0) || (N == ny - 1));
            if (PASS1 && !ring) {
                float FzT[KB + 1];                                 // flux through the face ABOVE level k0-1+h
#pragma unroll
                for (int h = 0; h <= KB; ++h) {
                    const float Wc = WN[h];
                    float f = upw(qN[h], qN[h + 1], Wc);          // at the top of the column q(h+1) == q(h): q*W (adv_mpdata.f90:96)
                    if (h == 0 && gnd) f = 0.f;                   // the ground
                    FzT[h] = f;
                }
                auto donor = [&](const int kk) {
                    const int h = kk + 1;
                    const float Uc = UN[kk], Vs = VN[kk], Vn = VNN[kk];
                    const float rdh = rdhN[kk], rdv = rdvN[kk];
                    const float FxL = upw(dpp_l(qN[h]), qN[h], Uc), FxR = dpp_r(FxL);
                    const float Fs = upw(qP[kk], qN[h], Vs), Fn = upw(qN[h], qNN[h], Vn);
                    const float v = qN[h] - ((FxR - FxL) + (Fn - Fs)) * rdh - (FzT[h] - FzT[h - 1]) * rdv;
                    q2N[h] = xring ? qN[h] : v;
                };
                // the two levels the neighbouring waves wait for go first and are posted before the others are computed
                donor(0);
                if (KB > 1) donor(KB - 1);
                s_q2[par][wv][0][lane] = q2N[1]; s_q2[par][wv][1][lane] = q2N[KB];
                MP_POST(0, seqA)
#pragma unroll
                for (int kk = 1; kk < KB - 1; ++kk) donor(kk);
            } else {
                // pass-1 field of the levels just below / above my own ones
                s_q2[par][wv][0][lane] = q2N[1]; s_q2[par][wv][1][lane] = q2N[KB];
                MP_POST(0, seqA)
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { mN[kk] = fmaxf(q2N[kk + 1], qN[kk + 1]); nN[kk] = fminf(q2N[kk + 1], qN[kk + 1]); }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {                      // what needs no neighbour, before the wait
                const int h = kk + 1;
                const float l = dpp_l(q2N[h]), r = dpp_r(q2N[h]);
                DxN[kk] = r - l; SxN[kk] = r + l;
                if (kk > 0 && kk < KB - 1) { DzN[kk] = q2N[h + 1] - q2N[h - 1]; SzN[kk] = q2N[h + 1] + q2N[h - 1]; }
            }
            MP_WAIT(0, seqA)
            q2N[0] = (wv > 0) ? s_q2[par][wv - 1][1][lane] : q2N[1];
            q2N[H - 1] = (wv < nw - 1) ? s_q2[par][wv + 1][0][lane] : q2N[KB];
            DzN[0] = q2N[2] - q2N[0]; SzN[0] = q2N[2] + q2N[0];
            if (KB > 1) { DzN[KB - 1] = q2N[KB + 1] - q2N[KB - 1]; SzN[KB - 1] = q2N[KB + 1] + q2N[KB - 1]; }
        } else {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { mN[kk] = nN[kk] = 0.f; DxN[kk] = SxN[kk] = DzN[kk] = SzN[kk] = 0.f; }
        }

        // The q part of the window rolls here, and the next plane of the scalar is requested now: three quarters of a
        // step cover its HBM latency instead of the one quarter left after the x/z limiter (1.41 -> 1.34 ms; a fourth
        // plane of q in registers, requested a whole step ahead, costs more in register pressure than it hides: 1.41).
        float qPh0n, qPh1n;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) qP[kk] = qN[kk + 1];
        qPh0n = qN[0]; qPh1n = qN[H - 1];
#pragma unroll
        for (int h = 0; h < H; ++h) qN[h] = qNN[h];
        __builtin_amdgcn_sched_barrier(0);
        ISSUE_LOADS_Q(qNN, P + 3)
        ISSUE_LOADS_Z(P)
        __builtin_amdgcn_sched_barrier(0);
        // 1 / (jaco rho), 1 / (jaco rho dz) of plane N: parked for the roll of the next step; those of plane P come back
        float rdhL[KB], rdvL[KB];
        {
            float pc[NC4 * 4];
#pragma unroll
            for (int t = 0; t < NC4; ++t) { const float4 v = s_park[NA4 + NB4 + t][tid]; pc[4 * t] = v.x; pc[4 * t + 1] = v.y; pc[4 * t + 2] = v.z; pc[4 * t + 3] = v.w; }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { rdhL[kk] = pc[kk]; rdvL[kk] = pc[KB + kk]; pc[kk] = rdhN[kk]; pc[KB + kk] = rdvN[kk]; }
#pragma unroll
            for (int t = 0; t < NC4; ++t) s_park[NA4 + NB4 + t][tid] = make_float4(pc[4 * t], pc[4 * t + 1], pc[4 * t + 2], pc[4 * t + 3]);
        }

        // ================= S2: y face between planes P and N =================
        float v2N[KB], FyN[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { v2N[kk] = 0.f; FyN[kk] = 0.f; }
        if (STEADY || (P >= 0 && haveN)) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int h = kk + 1;
                const float av = avN[kk], cvu = cvuN[kk], cvw = cvwN[kk];          // k_mpdata_coef
                const float t = av * (q2N[h] - q2P[h]) * frcp(q2N[h] + q2P[h] + EPSQ)
                              - cvu * (DxN[kk] + DxP[kk]) * frcp(SxN[kk] + SxP[kk] + EPSQ)
                              - cvw * (DzN[kk] + DzP[kk]) * frcp(SzN[kk] + SzP[kk] + EPSQ);
                v2N[kk] = t; FyN[kk] = upw(q2P[h], q2N[h], t);
            }
        }

        // ================= S3: x and z faces of plane P, their limiter, divergence =================
        float xdiv[KB], zdiv[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { xdiv[kk] = zdiv[kk] = 0.f; }
        float q2M[H];
        if (PARK) {
#pragma unroll
            for (int t = 0; t < NA4; ++t) { const float4 v = s_park[t][tid]; pkA[4 * t] = v.x; pkA[4 * t + 1] = v.y; pkA[4 * t + 2] = v.z; pkA[4 * t + 3] = v.w; }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) q2M[h] = pkA[h];
        const bool planeP = STEADY || ((P >= ja) && (P <= jb));                // warm-up planes feed nothing but the y limiter
        if (planeP) {
            float Dy[H], Sy[H];
#pragma unroll
            for (int h = 0; h < H; ++h) { Dy[h] = q2N[h] - q2M[h]; Sy[h] = q2N[h] + q2M[h]; }
            // ---- x faces (i-1/2) of my own levels
            float Fx[KB], u2[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int h = kk + 1;
                const float au = auP[kk], cuv = cuvP[kk], cuw = cuwP[kk];
                const float qL = dpp_l(q2P[h]);
                const float t = au * (q2P[h] - qL) * frcp(q2P[h] + qL + EPSQ)
                              - cuv * (Dy[h] + dpp_l(Dy[h])) * frcp(Sy[h] + dpp_l(Sy[h]) + EPSQ)
                              - cuw * (DzP[kk] + dpp_l(DzP[kk])) * frcp(SzP[kk] + dpp_l(SzP[kk]) + EPSQ);
                u2[kk] = t; Fx[kk] = upw(qL, q2P[h], t);
            }
            // ---- z faces above levels k0-1 .. k0+KB-1 (the lowest one is also computed by the wave below)
            float DxA[H], SxA[H];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { DxA[kk + 1] = DxP[kk]; SxA[kk + 1] = SxP[kk]; }
            { const float l0 = dpp_l(q2P[0]), r0 = dpp_r(q2P[0]), l1 = dpp_l(q2P[H - 1]), r1 = dpp_r(q2P[H - 1]);
              DxA[0] = r0 - l0; SxA[0] = r0 + l0; DxA[H - 1] = r1 - l1; SxA[H - 1] = r1 + l1; }
            float Fz[KB + 1], w2[KB + 1];
#pragma unroll
            for (int hf = 0; hf <= KB; ++hf) {
                const float aw = awP[hf], cwu = cwuP[hf], cwv = cwvP[hf];         // already times dz (:383-385); zero for the top level (:214)
                float t = aw * (q2P[hf + 1] - q2P[hf]) * frcp(q2P[hf + 1] + q2P[hf] + EPSQ)
                        - cwu * (DxA[hf] + DxA[hf + 1]) * frcp(SxA[hf] + SxA[hf + 1] + EPSQ)
                        - cwv * (Dy[hf] + Dy[hf + 1]) * frcp(Sy[hf] + Sy[hf + 1] + EPSQ);
                if (hf == 0 && gnd) t = 0.0f;                     // no face below the ground (slot 0 is a clamped load there)
                w2[hf] = t; Fz[hf] = upw(q2P[hf], q2P[hf + 1], t);
            }
            // ---- limiter, x direction
            float FxLim[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int h = kk + 1;
                if (FCT) {
                    const float qc = q2P[h], FxW = Fx[kk], FxE = dpp_r(Fx[kk]);
                    const float mL = dpp_l(mP[kk]), mR = dpp_r(mP[kk]), nL = dpp_l(nP[kk]), nR = dpp_r(nP[kk]);
                    float qmax = max3f(mL, mP[kk], mR), qmin = min3f(nL, nP[kk], nR);
                    if (xlo) { qmax = fmaxf(mP[kk], mR); qmin = fminf(nP[kk], nR); }
                    if (xhi) { qmax = fmaxf(mL, qc); qmin = fminf(nL, qc); }
                    float fin = fmaxf(0.f, FxW) - fminf(0.f, FxE), fout = fmaxf(0.f, FxE) - fminf(0.f, FxW);
                    if (xring) { fin = 0.f; fout = 0.f; }
                    const float bin = (qmax - qc) * frcp(fin + EPSF), bout = (qc - qmin) * frcp(fout + EPSF);
                    const float bLin = dpp_l(bin), bLout = dpp_l(bout);
                    const float s = fminf(1.0f, (u2[kk] > 0.0f) ? fminf(bin, bLout) : fminf(bLin, bout));
                    FxLim[kk] = s * Fx[kk];
                } else FxLim[kk] = Fx[kk];
                xdiv[kk] = dpp_r(FxLim[kk]) - FxLim[kk];
            }
            // ---- limiter, z direction
            float FzLim[KB + 1];
            if (FCT) {
                float bZin[H], bZout[H];
                auto betaz = [&](const int kk) {
                    const int h = kk + 1;
                    const float qc = q2P[h], FzB = Fz[h - 1], FzT = Fz[h];
                    const float mlo = (kk > 0) ? mP[kk - 1] : fmaxf(q2P[0], qPh0), nlo = (kk > 0) ? nP[kk - 1] : fminf(q2P[0], qPh0);
                    const float mhi = (kk < KB - 1) ? mP[kk + 1] : fmaxf(q2P[H - 1], qPh1), nhi = (kk < KB - 1) ? nP[kk + 1] : fminf(q2P[H - 1], qPh1);
                    float qmax = max3f(mlo, mP[kk], mhi), qmin = min3f(nlo, nP[kk], nhi);
                    float fin = fmaxf(0.f, FzB) - fminf(0.f, FzT), fout = fmaxf(0.f, FzT) - fminf(0.f, FzB);
                    // (level 0: mlo / nlo are the cell's own extrema and FzB = +-0, which is the reference's first-cell form)
                    if (h >= htop) { qmax = fmaxf(mlo, qc); qmin = fminf(nlo, qc); fin = fmaxf(0.f, FzB) - fminf(0.f, FzB); fout = fin; }
                    bZin[h] = (qmax - qc) * frcp(fin + EPSF); bZout[h] = (qc - qmin) * frcp(fout + EPSF);
                };
                betaz(0);                                          // the edge cells first: their betas are what the neighbouring waves wait for
                if (KB > 1) betaz(KB - 1);
                s_bz[par][wv][0][lane] = bZin[1]; s_bz[par][wv][1][lane] = bZout[1]; s_bz[par][wv][2][lane] = bZin[KB]; s_bz[par][wv][3][lane] = bZout[KB];
                MP_POST(1, seqB)
#pragma unroll
                for (int kk = 1; kk < KB - 1; ++kk) betaz(kk);
#pragma unroll
                for (int hf = 1; hf < KB; ++hf) {                  // face between slots hf (lower cell) and hf+1 (upper cell): own cells on both sides
                    const float s = fminf(1.0f, (w2[hf] > 0.0f) ? fminf(bZin[hf + 1], bZout[hf]) : fminf(bZin[hf], bZout[hf + 1]));
                    FzLim[hf] = s * Fz[hf];
                }
                MP_WAIT(1, seqB)
                bZin[0] = (wv > 0) ? s_bz[par][wv - 1][2][lane] : 0.f; bZout[0] = (wv > 0) ? s_bz[par][wv - 1][3][lane] : 0.f;
                bZin[H - 1] = (wv < nw - 1) ? s_bz[par][wv + 1][0][lane] : 0.f; bZout[H - 1] = (wv < nw - 1) ? s_bz[par][wv + 1][1][lane] : 0.f;
#pragma unroll
                for (int hf = 0; hf <= KB; hf += KB) {             // the two faces shared with a neighbouring wave
                    const float s = fminf(1.0f, (w2[hf] > 0.0f) ? fminf(bZin[hf + 1], bZout[hf]) : fminf(bZin[hf], bZout[hf + 1]));
                    FzLim[hf] = s * Fz[hf];
                }
            } else {
#pragma unroll
                for (int hf = 0; hf <= KB; ++hf) FzLim[hf] = Fz[hf];
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                zdiv[kk] = FzLim[kk + 1] - FzLim[kk];

            }
        }

        // ---- the inputs of the NEXT step: in flight while this step finishes (y limiter, store, roll) ----
        qPh0 = qPh0n; qPh1 = qPh1n;
        __builtin_amdgcn_sched_barrier(0);
        ISSUE_LOADS_A(P + 1)
        __builtin_amdgcn_sched_barrier(0);

        // ================= S4: beta_y of plane P ; S5: limited y face (P-1/2) =================
        float mM[KB], nM[KB], v2S[KB], FyS[KB], bYinM[KB], bYoutM[KB], acc[KB], rdhM[KB];
        if (PARK) {
#pragma unroll
            for (int t = 0; t < NB4; ++t) { const float4 v = s_park[NA4 + t][tid]; pkB[4 * t] = v.x; pkB[4 * t + 1] = v.y; pkB[4 * t + 2] = v.z; pkB[4 * t + 3] = v.w; }
        }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            mM[kk] = pkB[kk]; nM[kk] = pkB[KB + kk]; v2S[kk] = pkB[2 * KB + kk]; FyS[kk] = pkB[3 * KB + kk];
            bYinM[kk] = pkB[4 * KB + kk]; bYoutM[kk] = pkB[5 * KB + kk]; acc[kk] = pkB[6 * KB + kk]; rdhM[kk] = pkB[7 * KB + kk];
        }
        float bYin[KB], bYout[KB], FyLimS[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int h = kk + 1;
            if (FCT) {
                const float qc = q2P[h];
                float qmax = max3f(mM[kk], mP[kk], mN[kk]), qmin = min3f(nM[kk], nP[kk], nN[kk]);
                float fin = fmaxf(0.f, FyS[kk]) - fminf(0.f, FyN[kk]), fout = fmaxf(0.f, FyN[kk]) - fminf(0.f, FyS[kk]);
                if (!STEADY && P <= 0) { qmax = fmaxf(mP[kk], mN[kk]); qmin = fminf(nP[kk], nN[kk]); fin = 0.f; fout = 0.f; }
                if (!STEADY && P >= ny - 1) { qmax = fmaxf(mM[kk], qc); qmin = fminf(nM[kk], qc); fin = 0.f; fout = 0.f; }
                bYin[kk] = (qmax - qc) * frcp(fin + EPSF); bYout[kk] = (qc - qmin) * frcp(fout + EPSF);
                const float s = fminf(1.0f, (v2S[kk] > 0.0f) ? fminf(bYin[kk], bYoutM[kk]) : fminf(bYinM[kk], bYout[kk]));
                FyLimS[kk] = s * FyS[kk];
            } else { bYin[kk] = bYout[kk] = 0.f; FyLimS[kk] = FyS[kk]; }
        }

        // ================= S6: plane M is complete =================
        const int M = P - 1;
        if (STEADY || (M >= ja && M <= jb)) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const float v = xring ? q2M[kk + 1] : acc[kk] - FyLimS[kk] * rdhM[kk];
                if ((lane_out || (xring && lane < 64)) && (kk >= kst0 && kk <= kst1)) stb(out, bx, M * sj4 + kc[kk + 1], v);
            }
        }
        // the boundary rows of the new field are the old ones (adv_mpdata.f90:63-65)
        if (!STEADY && M == 0 && ja == 1) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
                if ((lane_out || xring) && (kk >= kst0 && kk <= kst1)) stb(out, bx, kc[kk + 1], q2M[kk + 1]);
        }
        if (!STEADY && P == ny - 1 && jb == ny - 2) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
                if ((lane_out || xring) && (kk >= kst0 && kk <= kst1)) stb(out, bx, (ny - 1) * sj4 + kc[kk + 1], q2P[kk + 1]);
        }
        // ---- roll the window
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int h = kk + 1;
            pkB[kk] = mP[kk]; pkB[KB + kk] = nP[kk]; pkB[2 * KB + kk] = v2N[kk]; pkB[3 * KB + kk] = FyN[kk];
            pkB[4 * KB + kk] = bYin[kk]; pkB[5 * KB + kk] = bYout[kk];
            pkB[6 * KB + kk] = q2P[h] - (xdiv[kk] - FyLimS[kk]) * rdhL[kk] - zdiv[kk] * rdvL[kk];
            pkB[7 * KB + kk] = rdhL[kk];
            mP[kk] = mN[kk]; nP[kk] = nN[kk];
            DxP[kk] = DxN[kk]; SxP[kk] = SxN[kk]; DzP[kk] = DzN[kk]; SzP[kk] = SzN[kk];
        }
#pragma unroll
        for (int h = 0; h < H; ++h) { pkA[h] = q2P[h]; q2P[h] = q2N[h]; }
        if (PARK) {
#pragma unroll
            for (int t = 0; t < NA4; ++t) s_park[t][tid] = make_float4(pkA[4 * t], pkA[4 * t + 1], pkA[4 * t + 2], pkA[4 * t + 3]);
#pragma unroll
            for (int t = 0; t < NB4; ++t) s_park[NA4 + t][tid] = make_float4(pkB[4 * t], pkB[4 * t + 1], pkB[4 * t + 2], pkB[4 * t + 3]);
        }

    };
    {
        const int s0 = ja + 1, s1 = max(min(jb, ny - 3), s0 - 1);  // steady steps: P in [s0, s1] (possibly empty)
        for (int ph = 0; ph < 2; ++ph) {                           // generic warm-up, steady bulk, generic tail
            const int lo = ph ? s1 + 1 : P0, hi = ph ? jb + 1 : s0 - 1;
            for (int P = lo; P <= hi; ++P) step(std::false_type{}, P);
            if (ph == 0)
                for (int P = s0; P <= s1; ++P) step(std::true_type{}, P);
        }
    }
#undef ISSUE_LOADS_A
#undef ISSUE_LOADS_Q
#undef ISSUE_LOADS_B
#undef LDP
#undef LDC
#undef ISSUE_LOADS_Z
#undef MP_POST
#undef MP_WAIT
#undef MP_WAIT1
#undef CLAMPJ
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int KB>
static void launch_fused(icar_hip_ctx *c, bool fct, bool pass1, const CVarPtrs &in, const VarPtrs &out, int nv,
                         int nw, int clen, int ntile, int nchunk, int nkr, int kstore)
{
    const unsigned nitem = (unsigned)(ntile * nchunk * nkr * nv), cap = (nitem + 7u) / 8u;
    const dim3 g(8u * cap), b(64, nw);                      // block id = xcd + 8 * slot, slot < cap
#define GO(F, P1) hipLaunchKernelGGL((k_mpdata_fused<KB, F, P1>), g, b, 0, c->stream, c->d, in, out, c->U, c->V, c->W, c->mpc, (int)(c->n3 * sizeof(float)), clen, ntile, nchunk, nv, nkr, kstore)
    if (fct) { if (pass1) GO(true, true); else GO(true, false); } else { if (pass1) GO(false, true); else GO(false, false); }
#undef GO
}

// the scalar-independent coefficients of this step's Courant winds (c->U, c->V, c->Wdz), for icar_hip_setup_winds
int icar_mpdata_coef_run(icar_hip_ctx *c, bool rho_on)
{
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    const float *rho = rho_on ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    if (!jaco || !dz || (rho_on && !rho)) return 1;
    if (c->n3 * sizeof(float) * MPC_N >= ((size_t)1 << 31)) { icar_set_error("mpdata: a tile of more than 48 M cells is not supported (32-bit buffer offsets into the coefficient arrays)"); return 1; }
    if (!c->mpc) HIPCHK(hipMalloc(&c->mpc, c->n3 * sizeof(float) * MPC_N));
    const dim3 g((c->d.nx + 63) / 64, (c->d.nz + 3) / 4, c->d.ny), b(64, 4);
    if (rho_on) hipLaunchKernelGGL(k_mpdata_coef<true>, g, b, 0, c->stream, c->d, c->U, c->V, c->Wdz, rho, jaco, dz, c->mpc);
    else        hipLaunchKernelGGL(k_mpdata_coef<false>, g, b, 0, c->stream, c->d, c->U, c->V, c->Wdz, rho, jaco, dz, c->mpc);
    HIPCHK(hipGetLastError());
    c->mpc_dens = rho_on ? 1 : 0;
    return 0;
}

// one corrective iteration of MPDATA: in -> out (distinct buffers).  pass1: the donor-cell pass is part of it (iord == 2).
int icar_mpdata_fused_run(icar_hip_ctx *c, bool rho_on, bool fct, bool pass1, const CVarPtrs &in, const VarPtrs &out, int nv)
{
    const int nx = c->d.nx, nz = c->d.nz, ny = c->d.ny;
    if (!c->mpc || c->mpc_dens != (rho_on ? 1 : 0)) { icar_set_error("mpdata: icar_hip_setup_winds(scheme 2) with the same advect_density must come first"); return 1; }
    if (nx < 3 || ny < 3) { icar_set_error("mpdata: tile must be at least 3 x 3 cells"); return 1; }
    if ((size_t)nx * nz * ny * sizeof(float) >= ((size_t)1 << 31)) { icar_set_error("mpdata: a field of 2 GiB or more is not supported (32-bit buffer offsets)"); return 1; }
    const int ntile = std::max(1, (nx - 2 + MP_XOUT - 1) / MP_XOUT);
    // levels: one block holds at most MP_NW x MP_KB = 40; taller columns are cut into level ranges with MP_ZH halo levels
    const int cap_lv = MP_NW * MP_KB;
    int nkr = 1;
    while (nkr * cap_lv - 2 * MP_ZH * (nkr - 1) < nz) ++nkr;
    const int kstore = (nz + nkr - 1) / nkr;
    const int blk_lv = std::min(nz, kstore + (nkr > 1 ? 2 * MP_ZH : 0));       // levels a block computes
    const int kb = (blk_lv + MP_NW - 1) / MP_NW, nw = (blk_lv + kb - 1) / kb;
    // y chunks: a block keeps a whole CU, so the work items (tiles x chunks x ranges x scalars) should fill a whole number of
    // rounds of the 256 CUs; each chunk pays ~3.5 warm-up planes.  Pick the chunk count with the best product of the two.
    const int rows = ny - 2, per = ntile * nkr * nv;
    int nchunk = 1; double best = -1.0;
    for (int n = 1; n <= std::max(1, rows / 16); ++n) {
        const int cl = (rows + n - 1) / n, nn = (rows + cl - 1) / cl;
        const double items = (double)per * nn, rounds = std::ceil(items / 256.0);
        const double eff = items / (256.0 * rounds) * cl / (cl + 3.5);
        if (eff > best + 1e-9) { best = eff; nchunk = nn; }
    }
    const int clen = (rows + nchunk - 1) / nchunk;
    nchunk = (rows + clen - 1) / clen;
#define KBCASE(K) case K: launch_fused<K>(c, fct, pass1, in, out, nv, nw, clen, ntile, nchunk, nkr, kstore); break;
    switch (kb) { KBCASE(1) KBCASE(2) KBCASE(3) KBCASE(4) KBCASE(5) default: icar_set_error("mpdata: internal level-range error"); return 1; }
#undef KBCASE
    HIPCHK(hipGetLastError());
    return 0;
}
