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
//     U * (U > 0 ? l : r), which is the same number.  Results agree with the CPU reference to <= 8.1e-7 of the local field
//     scale (profiles/r06_parity.json); tests assert 1e-5 on every cell (north-star tolerance) and fail above 0.3 of it -- the donor-cell kernel of the upwind scheme
//     (advect.hip) stays bit-exact.
//
// Everything that does not depend on the scalar is computed ONCE per step by k_mpdata_coef (icar_hip_setup_winds) and loaded:
// per face the antidiffusive coefficient |U|(1-|U|/Gbar)/2 and the two cross-term factors U Ubar_perp / (8 Gbar) (the six 4-point
// transverse Courant averages, the 1/(G_i + G_i-1), the ground / top / x-ring zeros folded in, the z faces already times dz), and
// per cell the two denominators jaco rho, dz jaco rho of the donor-cell passes -- eleven arrays (round 2 recomputed all of it for each
// of the 9 scalars: ~50 of 252 VALU instructions per scalar-cell).  They and U_m, V_m, W_m are re-read per scalar from L2 (60 B per
// cell against the 8 B of the scalar itself), which is why blocks of the same (tile, chunk) and different scalars are scheduled
// onto the same XCD.
//
// Round 6: what the kernel is bound by is its LOADS IN FLIGHT, not its arithmetic (at two waves per SIMD ~1 % of its time per load of
// a step's ~80; eight more VALU instructions per cell measured as nothing, profiles/r06_steps.md).  Hence: the donor-cell pass divides
// exactly (bit-identical q2, see exact_quot) with computed reciprocals, and the plane's final update one step later takes those
// reciprocals from it (registers / thread-private LDS) instead of loading them -- 81 loads per step where round 5 had 91.
#include "ctx.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

#define MP_HL 3                       // halo lanes per side
#define MP_XOUT (64 - 2 * MP_HL)      // outputs per 64-lane tile
#define MP_NW 8                       // waves per block at most: two per SIMD, ~250 VGPRs each (12 x 3 levels at three per SIMD: no faster, profiles/r05_steps.md)
#define MP_KB 5                       // levels per thread at most
#define MP_ZH 2                       // halo levels of a level range: a fake edge corrupts the outputs of the 2 levels next to it
#define EPSQ 1e-10f
#define EPSF 1e-15f
#define HEPSQ 0.5e-10f                // every cross-term denominator is the sum of TWO stencil sums + EPSQ: each sum carries half of it

// bound_ctrl:1 -- a lane without a source reads 0, so no `old` value has to be materialised and the shift can fold into
// the consuming VALU instruction (v_sub_f32_dpp ...)
__device__ __forceinline__ float dpp_l(float x)   // value of lane-1 (0 in lane 0)
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_r(float x)   // value of lane+1 (0 in lane 63)
{ return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true)); }
// the same shifts with a dummy `old` operand: a DPP read folds into its consumer only when it has ONE consumer, and two reads of
// one register written with dpp_l / dpp_r are merged into one v_mov_b32_dpp with two users.  (bound_ctrl:1: `old` is never read.)
__device__ __forceinline__ float dpp_l2(float x, float o)
{ return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(o), __float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_r2(float x, float o)
{ return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(o), __float_as_int(x), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }   // keeps max(max(a, b), c) two DPP-foldable v_max instead of v_max3 + two moves
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float upw(float l, float r, float U) { return U * (U > 0.0f ? l : r); }      // == flux1 (adv_mpdata.f90:40)
// a / g rounded ONCE, without an IEEE division and without a second array: r0 = v_rcp_f32(g) (1 ulp), one Newton step
// r = r0 + r0 (1 - g r0), then y = RN(a r) (within 2 ulp of the quotient), the residual e = a - g y (exact in an fma) and
// RN(y + e r) = a / g + (a / g - y) O(2^-23): the correctly rounded quotient unless a / g lies within ~2^-22 ulp of a rounding
// boundary.  On the CPU, 2e10 random (a, g) with r0 off by up to 3 ulp: 0 differences from a / g (without the Newton step: 1800 at
// 1 ulp; with two correction steps instead: 350) -- profiles/r06_steps.md.  The donor-cell pass divides this way: its result q2
// decides the limiter's all-or-nothing factors next to the ring (see there).  Round 6's first form loaded RN(1 / g) next to g: the
// two extra loads per level cost the kernel 9 %, the arithmetic nothing (it is bound by the loads in flight, not by the VALU).
__device__ __forceinline__ float exact_quot(float a, float g, float &r)          // r: the refined reciprocal, for whoever divides by g again
{
    const float r0 = frcp(g);
    r = __builtin_fmaf(__builtin_fmaf(-g, r0, 1.0f), r0, r0);
    const float y = opaque(a * r);
    return __builtin_fmaf(__builtin_fmaf(-g, y, a), r, y);
}
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
//   cell:                                  Gh = jaco rho, Gv = (dz jaco) rho -- the denominators of the donor-cell passes in the
//                                          reference's association (adv_mpdata.f90:86-99): exact_quot() in k_mpdata_fused
// a? = |C| (1 - 2 |C| / (G + G')) / 2 ;  c?? = C (sum of the 4 transverse Courant numbers around the face) / (16 (G + G'))
// with G = jaco [rho]; cross terms through the ground / column top (k-1, k+1 missing) and in the x ring are zero.
// ------------------------------------------------------------------------------------------------
// (enum MPC_*: ctx.h)
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
    // (ring cells keep their value, adv_mpdata.f90:63-65: k_mpdata_fused multiplies their flux differences by zero)
    const float gh = g, gv = RHO ? (dzc * jaco[c]) * rho[c] : dzc * jaco[c];        // (dz * jaco * rho is evaluated left to right)
    C[MPC_GH * n3 + c] = gh; C[MPC_GV * n3 + c] = gv;
}

// mode of one step of the march: generic (any plane, rolls by copying) or one half of a steady pair (its exchange-buffer parity)
template <bool ST, int PAR> struct MpMode { static constexpr bool steady = ST; static constexpr int par = PAR; };

// KB levels per thread (at most MP_NW waves per block), FCT: limiter on (advect_density lives in the coefficients),
// PASS1: first corrective iteration (donor-cell pass inside); false: iord >= 3, where q2 == q (adv_mpdata.f90:393-402),
// EXACT: the block's waves hold exactly the column (nz == waves x KB, one level range): the top of the column is slot KB of the
// last wave, known at compile time, and every slot is stored -- no per-slot level tests at all.
//
// Round 5: the step is written for the ISSUE rate of a wave, not for the VALU pipe.  At two waves per SIMD one wave issues an
// instruction only every 6.5-8 clocks whatever its kind (profiles/micro/issuebench.hip: s_add_i32 costs what v_add_f32 costs, a
// not-taken branch pair 17 clocks, s_nop 7), and the round-4 steady step was 1103 VALU + 556 SALU + 150 other instructions in
// ~100 basic blocks.  What changed:
//   * boundary forms come out of the DATA: 1/(jaco rho) and 1/(jaco rho dz) are zero on the ring cells (k_mpdata_coef), so the
//     donor-cell pass and the final update return q there without a select; lanes left / right of the domain are clamped copies
//     of the ring column, so the limiter's first / last-cell extrema are the general three-cell form; the ring's zero in / out
//     flow is a per-lane factor folded into the fma that adds the epsilon; the ground is a wave-uniform factor on two values;
//     neighbour waves that do not exist are the wave itself (address selection, once).  The steady step has no branch left but
//     the four spin waits and one store mask.
//   * buffer offsets: one VGPR per level slot (loop-invariant) + one scalar per (array, plane): 17 s_add per step instead of 98.
//   * the window does not roll: the steady loop runs PAIRS of steps with the two register sets of each rolling quantity
//     swapped (q: planes N / NN; q2, extrema, stencils: planes P / N), and the next plane of the scalar is loaded straight into
//     the set that has just been used up.  The donor-cell flux through a y face is carried to the next step (it was computed
//     twice), which also makes the q window two planes deep and drops a plane of V loads.
template <int KB, bool FCT, bool PASS1, bool EXACT>
// 248 VGPRs, not the 256 two waves per SIMD could have (the attribute counts half of gfx90a+'s unified file: 124 -> 248).  The
// 248 blocks of a launch hold their CUs for the whole kernel, so whatever the host issues on the second stream beside the
// advection (whole-field forcing, the CFL reduction of the next update_dt) can only run in what these waves leave: with 2 x 248
// of a SIMD's 512 registers taken, one more wave of <= 16 VGPRs fits and those streaming kernels run concurrently; at 254 (what
// the allocator takes if allowed) they wait for the launch to end -- the advection alone is then 5 % faster, the step 2 % slower.
__global__ void __launch_bounds__(64 * MP_NW) __attribute__((amdgpu_num_vgpr(124)))
k_mpdata_fused(Dims d, CVarPtrs qin, VarPtrs qout,
               const float *__restrict__ Ug, const float *__restrict__ Vg, const float *__restrict__ Wg,
               const float *__restrict__ Cg, unsigned asz, int clen, int ntile, int nchunk, int nscal, int nkr, int kstore)
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
    // below and the wave above it, nothing else.  A wave posts its edges (data, then a step counter stored with release semantics; the
    // reader's acquire load of the counter makes the data visible to it), does the work that needs no neighbour, and
    // spins on the two neighbouring counters.  The double buffering by step parity stays sufficient: wave w overwrites a buffer
    // at step t + 2 only after it has seen the counters of w-1 / w+1 at t + 1, which they post after their reads of step t.
    // Counters: s_sync[e * NW2 + 1 + wave]; the entries below wave 0 and above the last wave are INT_MAX (never waited for).
    // Every lane stores (no EXEC change): lane 0 to the counter, the others to a dump area behind the counters.
    constexpr int NW2 = MP_NW + 2;
    __shared__ int s_sync[3 * NW2 + 64];
    constexpr int NA4 = (H + 3) / 4, NB4 = 2 * KB;  // float4 slots: q2M[H] | per level {mM nM v2S FyS} {bYinM bYoutM acc rdhM}
    __shared__ float4 s_park[NA4 + NB4][64 * MP_NW];
    // 1 / (dz jaco rho) of plane N, from the donor-cell pass to the plane's final update one step later (thread-private, by step parity:
    // no synchronisation).  Re-reading dz jaco rho there was 5 more loads in flight per step; 1 / (jaco rho) travels in registers.
    constexpr int NG4 = (KB + 3) / 4;
    __shared__ float4 s_rv[2][NG4][64 * MP_NW];

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
        for (int t = tid; t < 2 * NW2; t += 64 * nw) { const int j = t % NW2; s_sync[t] = (j == 0 || j > nw) ? 0x7fffffff : 0; }
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
    const int ic = min(max(i, 0), nx - 1);                  // lanes outside the domain are copies of the ring column
    const bool xin = (i > 0) && (i < nx - 1), xring = (i == 0) || (i == nx - 1);
    const bool lane_store = ((lane >= MP_HL) && (lane < 64 - MP_HL) && xin) || xring;
    // the limiter sees no flow into or out of a ring cell, and the donor-cell pass leaves it alone; the lanes beyond the ring are
    // copies of the ring column in this too
    float rm = (i <= 0 || i >= nx - 1) ? 0.0f : 1.0f;
    const int ja = 1 + chunk * clen, jb = min(ja + clen - 1, ny - 2);
    const int k0 = kbase + wv * KB;
    // byte offset of (column, level of slot h) -- slot h (0..H-1) = level k0-1+h, clamped: the slot above the top level holds the
    // top level's own values and the slot below level 0 those of level 0: upw(q(top), q(top+1), W) IS q(top) W
    // (adv_mpdata.f90:96), and the z limiter's first-cell form (extrema without the cell below, no flux through the ground)
    // falls out of the general one because max(q2(0), l(0)) is the cell's own extremum and the ground flux is +-0.
    int vk[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { vk[h] = 4 * ic + min(max(k0 - 1 + h, 0), nz - 1) * nx * 4; asm volatile("" : "+v"(vk[h])); }
    asm volatile("" : "+v"(rm));
    // nothing flows through the ground (slot 0 is a clamped load there): a wave-uniform factor
    const float gmul = __int_as_float(__builtin_amdgcn_readfirstlane((k0 == 0) ? 0 : 0x3f800000));
    // what is left of the level flags: the last-cell form of the z limiter at the top of the column and, unless EXACT, the
    // store range.  EXACT: slot KB of the last wave, one wave-uniform flag.  Otherwise three scalars, kept opaque so that the
    // comparisons are redone by the scalar unit inside the step instead of living in SGPR pairs across it.
    const bool topwave = (k0 + KB == nz);
    const int htop_u = __builtin_amdgcn_readfirstlane(nz - k0), kst0_u = __builtin_amdgcn_readfirstlane(ka - k0), kst1_u = __builtin_amdgcn_readfirstlane(kb - k0);
    // neighbour waves: where the wave below / above does not exist the wave reads its own edge instead (for the pass-1 field
    // that IS the clamped halo level; the betas read this way limit a face whose flux is zero)
    const int wlo = max(wv - 1, 0), whi = min(wv + 1, nw - 1);
    const int q2lo = (wv > 0) ? 1 : 0, q2hi = (wv < nw - 1) ? 0 : 1;          // which edge of that wave
    const int bzlo = (wv > 0) ? 2 : 0, bzhi = (wv < nw - 1) ? 0 : 2;
    int *const fpost = (lane == 0) ? &s_sync[1 + wv] : &s_sync[2 * NW2 + lane];
    int seqA = 0, seqB = 0;
// Flag store = release, flag load = acquire, both at workgroup scope: the data written to s_q2 / s_bz before a post is visible to
// the wave that has seen the counter -- by the memory model, not by what the LDS happens to do today.  In the
// non-tgsplit mode these lower to the same ds_write / ds_read plus an s_waitcnt lgkmcnt(0) in front of the store.
#define MP_POST(e, seq)                                                                                                   \
    {                                                                                                                      \
        ++seq;                                                                                                             \
        __hip_atomic_store(fpost + (e) * NW2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);                         \
    }
#define MP_WAIT(e, seq)                                                                                                    \
    {                                                                                                                      \
        int spin = 0;                                                                                                      \
        for (;;) {                                                                                                         \
            const int fa = __hip_atomic_load(&s_sync[(e) * NW2 + wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);      \
            const int fb = __hip_atomic_load(&s_sync[(e) * NW2 + wv + 2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);  \
            if (__builtin_amdgcn_readfirstlane(min(fa, fb)) >= seq) break;                                                 \
            __builtin_amdgcn_s_sleep(1);                                                                                   \
            if (++spin > (1 << 26)) __builtin_trap();              /* a lost neighbour must not hang the device */          \
        }                                                                                                                  \
    }

#define LDQ(h, po) ldb(q, vk[h], (po))
#define LDC(a, h, po) ldb(cr, vk[h], (int)((unsigned)(a) * asz + (unsigned)(po)))   /* coefficient array a (MPC_*); unsigned: the eleven arrays may span up to 4 GiB */
#define CLAMPJ(p) min(max((p), 0), ny - 1)

    // ---- rolling state (planes relative to the step's in-plane index P; N = P+1, M = P-1) ----
    struct QBuf { float v[H]; };                            // one plane of the scalar (= l of the limiter)
    struct PSet {                                           // everything of a plane that a later step reads from registers
        float q2[H];                                        // field after pass 1
        float m[KB], n[KB];                                 // max / min of (q2, l) per cell
        float Dx[KB], Sx[KB], Dz[KB], Sz[KB];               // q2(i+1) -+ q2(i-1), q2(k+1) -+ q2(k-1)
        float Fyd[KB];                                      // donor-cell flux through the plane's NORTH face (pass 1)
        float rh[KB];                                       // 1 / (jaco rho) as the donor-cell pass refined it, kept for the plane's final update
        float mh0, nh0, mh1, nh1;                           // extrema of the two halo levels (z limiter)
    };
    QBuf Q0, Q1;
    PSet S0, S1;
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) { S0.m[kk] = S0.n[kk] = 0.f; S0.Dx[kk] = S0.Sx[kk] = S0.Dz[kk] = S0.Sz[kk] = 0.f; S0.Fyd[kk] = 0.f; S0.rh[kk] = 0.f; }
#pragma unroll
    for (int h = 0; h < H; ++h) S0.q2[h] = 0.f;
    S0.mh0 = 0.f; S0.nh0 = 0.f; S0.mh1 = 0.f; S0.nh1 = 0.f;
    S1 = S0;
#pragma unroll
    for (int t = 0; t < NA4 + NB4; ++t) s_park[t][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 2 * NG4; ++t) s_rv[t / NG4][t % NG4][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int P0 = ja - 3;
    // inputs of one step that are requested during the step before it
    float WN[KB + 1], UN[KB], VNN[KB], ghN[KB], gvN[KB];
// group A: what the donor-cell pass of step PP reads (plane PP+1; the north face's V on plane PP+2)
#define ISSUE_LOADS_A(oN_, oNN_)                                                                                         \
    {                                                                                                                    \
        _Pragma("unroll") for (int h = 0; h <= KB; ++h) WN[h] = ldb(Wr, vk[h], (oN_));                                   \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { UN[kk] = ldb(Ur, vk[kk + 1], (oN_)); VNN[kk] = ldb(Vr, vk[kk + 1], (oNN_)); } \
        _Pragma("unroll") for (int kk = 0; kk < KB; ++kk) { ghN[kk] = LDC(MPC_GH, kk + 1, (oN_)); gvN[kk] = LDC(MPC_GV, kk + 1, (oN_)); } \
    }
    {
        const int o0 = CLAMPJ(P0) * sj4, o1 = CLAMPJ(P0 + 1) * sj4, o2 = CLAMPJ(P0 + 2) * sj4;
        float qP0[KB], V1[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { qP0[kk] = LDQ(kk + 1, o0); V1[kk] = ldb(Vr, vk[kk + 1], o1); }
#pragma unroll
        for (int h = 0; h < H; ++h) Q0.v[h] = LDQ(h, o1);
#pragma unroll
        for (int h = 0; h < H; ++h) Q1.v[h] = LDQ(h, o2);
        ISSUE_LOADS_A(o1, o2)
        if (PASS1) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) S0.Fyd[kk] = opaque(upw(qP0[kk], Q0.v[kk + 1], V1[kk]));     // the face between planes P0 and P0+1
        }
    }
    // One step of the march.  STEADY = every stage is on and no plane of the window is a boundary row of the domain (the bulk of
    // a chunk): qN / qNN and sP / sN are the two register sets in this step's roles, nothing is copied, PAR is a constant.
    // The generic form runs the warm-up steps of a chunk, its last steps and the chunks that touch row 0 / ny-1, always with
    // the sets in the roles (Q0, Q1, S0, S1), and rolls them by copying.
    auto step = [&](auto md, QBuf &qN, QBuf &qNN, PSet &sP, PSet &sN, const int P) {
        constexpr bool STEADY = decltype(md)::steady;
        const int par = STEADY ? decltype(md)::par : ((P - P0) & 1);
        const int N = P + 1;
        const bool haveN = STEADY || ((N >= 0) && (N <= ny - 1));
        const int oP = (STEADY ? P : CLAMPJ(P)) * sj4, oN = (STEADY ? N : CLAMPJ(N)) * sj4;
        int htop = htop_u, kst0 = kst0_u, kst1 = kst1_u;
        if (!EXACT) asm volatile("" : "+s"(htop), "+s"(kst0), "+s"(kst1));
        // Group A of this step's global loads was issued during the previous step (after its x/z limiter), group B -- the x / y
        // face coefficients -- is issued here; each group back to back: loads placed next to their use were waited for one by one.
        float avN[KB], cvuN[KB], cvwN[KB], auP[KB], cuvP[KB], cuwP[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { avN[kk] = LDC(MPC_AV, kk + 1, oN); cvuN[kk] = LDC(MPC_CVU, kk + 1, oN); cvwN[kk] = LDC(MPC_CVW, kk + 1, oN); }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { auP[kk] = LDC(MPC_AU, kk + 1, oP); cuvP[kk] = LDC(MPC_CUV, kk + 1, oP); cuwP[kk] = LDC(MPC_CUW, kk + 1, oP); }
        __builtin_amdgcn_sched_barrier(0);

        // ================= S1: donor-cell pass on plane N, its extrema and x/z differences =================
        float *const q2N = sN.q2;
        const float *const q2P = sP.q2;
        // S2, the y face between planes P and N, level by level
        float v2N[KB], FyN[KB], rvN[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { v2N[kk] = 0.f; FyN[kk] = 0.f; rvN[kk] = 0.f; }
        auto yface = [&](const int kk) {
            const int h = kk + 1;
            const float t = avN[kk] * (q2N[h] - q2P[h]) * frcp(q2N[h] + q2P[h] + EPSQ)
                          - cvuN[kk] * (sN.Dx[kk] + sP.Dx[kk]) * frcp(sN.Sx[kk] + sP.Sx[kk])       // (the sums carry EPSQ / 2 each)
                          - cvwN[kk] * (sN.Dz[kk] + sP.Dz[kk]) * frcp(sN.Sz[kk] + sP.Sz[kk]);
            v2N[kk] = t; FyN[kk] = upw(q2P[h], q2N[h], t);
        };
        if (haveN) {
            if (PASS1) {
                const float rmN = (STEADY || (N > 0 && N < ny - 1)) ? rm : 0.0f;
                float FzT[KB + 1];                                 // flux through the face ABOVE level k0-1+h
#pragma unroll
                for (int h = 0; h <= KB; ++h) FzT[h] = opaque(upw(qN.v[h], qN.v[h + 1], WN[h]));   // at the top of the column q(h+1) == q(h): q*W (adv_mpdata.f90:96)
                FzT[0] *= gmul;                                    // the ground
                auto donor = [&](const int kk) {
                    const int h = kk + 1;
                    // q2 is BIT-IDENTICAL to the reference's (adv_mpdata.f90:66-99: same fluxes, same association, both quotients
                    // rounded once, no contraction).  It has to be: next to the ring the limiter's factor is (q2 - qmin) / 1e-15 or
                    // (qmax - q2) / 1e-15 clipped to 1 (adv_mpdata_FCT_core.f90:80-113 with fin = fout = 0 there) -- zero or one,
                    // decided by whether q2 of the ring's neighbour EQUALS an extremum: one ulp in q2 switches a whole antidiffusive
                    // flux on or off (round 5's contracted form: 9.2e-6 of theta at one cell of the config-3 tile).  Everything
                    // downstream of q2 is continuous in its rounding errors.
                    const float FxL = opaque(upw(dpp_l(qN.v[h]), qN.v[h], UN[kk])), FxR = dpp_r(FxL);
                    const float Fn = opaque(upw(qN.v[h], qNN.v[h], VNN[kk]));
                    // ring cells keep their value (adv_mpdata.f90:63-65): the flux differences times 0 (rmN: the x ring's lanes, and
                    // every lane when plane N is row 0 or ny-1 -- generic steps only)
                    const float dh = ((FxR - FxL) + (Fn - sP.Fyd[kk])) * rmN, dv = (FzT[h] - FzT[h - 1]) * rmN;
                    const float t = qN.v[h] - exact_quot(dh, ghN[kk], sN.rh[kk]);       // (the reciprocals serve the plane's final update too)
                    q2N[h] = t - exact_quot(dv, gvN[kk], rvN[kk]);
                    sN.Fyd[kk] = Fn;
                };
                // the two levels the neighbouring waves wait for go first and are posted before the others are computed
                donor(0);
                if (KB > 1) donor(KB - 1);
                s_q2[par][wv][0][lane] = q2N[1]; s_q2[par][wv][1][lane] = q2N[KB];
                MP_POST(0, seqA)
#pragma unroll
                for (int kk = 1; kk < KB - 1; ++kk) donor(kk);
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) q2N[h] = qN.v[h];      // iord >= 3: q2 == q, halo levels included (no exchange)
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) { sN.rh[kk] = frcp(ghN[kk]); rvN[kk] = frcp(gvN[kk]); }
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { sN.m[kk] = fmaxf(q2N[kk + 1], qN.v[kk + 1]); sN.n[kk] = fminf(q2N[kk + 1], qN.v[kk + 1]); }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {                      // what needs no neighbour, before the wait
                const int h = kk + 1;
                const float l = dpp_l(q2N[h]);
                sN.Dx[kk] = dpp_r(q2N[h]) - l; sN.Sx[kk] = dpp_r2(q2N[h], l) + (l + HEPSQ);
                if (kk > 0 && kk < KB - 1) { sN.Dz[kk] = q2N[h + 1] - q2N[h - 1]; sN.Sz[kk] = q2N[h + 1] + (q2N[h - 1] + HEPSQ); }
            }
            if (PASS1) {
                MP_WAIT(0, seqA)
                q2N[0] = s_q2[par][wlo][q2lo][lane];
                q2N[H - 1] = s_q2[par][whi][q2hi][lane];
            }
            sN.Dz[0] = q2N[2] - q2N[0]; sN.Sz[0] = q2N[2] + (q2N[0] + HEPSQ);
            if (KB > 1) { sN.Dz[KB - 1] = q2N[KB + 1] - q2N[KB - 1]; sN.Sz[KB - 1] = q2N[KB + 1] + (q2N[KB - 1] + HEPSQ); }
            sN.mh0 = fmaxf(q2N[0], qN.v[0]); sN.nh0 = fminf(q2N[0], qN.v[0]);
            sN.mh1 = fmaxf(q2N[H - 1], qN.v[H - 1]); sN.nh1 = fminf(q2N[H - 1], qN.v[H - 1]);
        } else {
#pragma unroll
            for (int h = 0; h < H; ++h) q2N[h] = qN.v[h];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { sN.m[kk] = sN.n[kk] = 0.f; sN.Dx[kk] = sN.Sx[kk] = sN.Dz[kk] = sN.Sz[kk] = 0.f; sN.Fyd[kk] = 0.f; sN.rh[kk] = 0.f; }
            sN.mh0 = 0.f; sN.nh0 = 0.f; sN.mh1 = 0.f; sN.nh1 = 0.f;
        }

        // 1 / (dz jaco rho) of plane N for its final update in the next step
        {
            float pv[NG4 * 4];
#pragma unroll
            for (int t = 0; t < NG4 * 4; ++t) pv[t] = (t < KB) ? rvN[t < KB ? t : 0] : 0.f;
#pragma unroll
            for (int t = 0; t < NG4; ++t) s_rv[par][t][tid] = make_float4(pv[4 * t], pv[4 * t + 1], pv[4 * t + 2], pv[4 * t + 3]);
        }
        // Plane N of the scalar has been used up: the next plane is requested now, three quarters of a step ahead (it is the
        // one input that ALWAYS comes from HBM) -- in the steady loop straight into the registers of plane N.
        __builtin_amdgcn_sched_barrier(0);
        // group Z: the z face coefficients of plane P -- requested BEFORE the scalar:
        // vmcnt counts in order, so cache-resident loads issued behind an HBM load are waited for as long as that one
        float awP[KB + 1], cwuP[KB + 1], cwvP[KB + 1];
#pragma unroll
        for (int h = 0; h <= KB; ++h) { awP[h] = LDC(MPC_AW, h, oP); cwuP[h] = LDC(MPC_CWU, h, oP); cwvP[h] = LDC(MPC_CWV, h, oP); }
        {
            const int o3 = (STEADY ? P + 3 : CLAMPJ(P + 3)) * sj4;
            if (STEADY) {
#pragma unroll
                for (int h = 0; h < H; ++h) qN.v[h] = LDQ(h, o3);
            } else {
                qN = qNN;
#pragma unroll
                for (int h = 0; h < H; ++h) qNN.v[h] = LDQ(h, o3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= S2: y face between planes P and N =================
        if (STEADY || (P >= 0 && haveN)) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) yface(kk);
        }

        // ================= S4: beta_y of plane P ; S5: limited y face (P-1/2) =================
        // parked by the step before: per level {mM nM v2S FyS} {bYinM bYoutM acc rdhM} -- extrema of plane M; pseudo-velocity /
        // unlimited flux of the y face (P-1/2); beta_y of plane M; q2 - x/z/south contributions of plane M; its 1 / (jaco rho).
        float bYin[KB], bYout[KB], FyLimS[KB], outM[KB];
        auto ylim = [&]() {
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int h = kk + 1;
            const float4 pa = s_park[NA4 + 2 * kk][tid], pb = s_park[NA4 + 2 * kk + 1][tid];
            const float mM = pa.x, nM = pa.y, v2S = pa.z, FyS = pa.w, bYinM = pb.x, bYoutM = pb.y, acc = pb.z, rdhM = pb.w;
            if (FCT) {
                const float qc = q2P[h];
                float qmax = max3f(mM, sP.m[kk], sN.m[kk]), qmin = min3f(nM, sP.n[kk], sN.n[kk]);
                float fin = fmaxf(0.f, FyS) - fminf(0.f, FyN[kk]), fout = fmaxf(0.f, FyN[kk]) - fminf(0.f, FyS);
                if (!STEADY && P <= 0) { qmax = fmaxf(sP.m[kk], sN.m[kk]); qmin = fminf(sP.n[kk], sN.n[kk]); fin = 0.f; fout = 0.f; }
                if (!STEADY && P >= ny - 1) { qmax = fmaxf(mM, qc); qmin = fminf(nM, qc); fin = 0.f; fout = 0.f; }
                bYin[kk] = (qmax - qc) * frcp(fin + EPSF); bYout[kk] = (qc - qmin) * frcp(fout + EPSF);
                const float s = fminf(1.0f, (v2S > 0.0f) ? fminf(bYin[kk], bYoutM) : fminf(bYinM, bYout[kk]));
                FyLimS[kk] = s * FyS;
            } else { bYin[kk] = bYout[kk] = 0.f; FyLimS[kk] = FyS; }
            outM[kk] = acc - FyLimS[kk] * rdhM;                    // plane M is complete (ring cells: rdh = 0, acc = q)
        }
        };
        // ================= S3: x and z faces of plane P, their limiter, divergence =================
        float xdiv[KB], zdiv[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) { xdiv[kk] = zdiv[kk] = 0.f; }
        float q2M[NA4 * 4];
#pragma unroll
        for (int t = 0; t < NA4; ++t) { const float4 v = s_park[t][tid]; q2M[4 * t] = v.x; q2M[4 * t + 1] = v.y; q2M[4 * t + 2] = v.z; q2M[4 * t + 3] = v.w; }
        const bool planeP = STEADY || ((P >= ja) && (P <= jb));                // warm-up planes feed nothing but the y limiter
        if (planeP) {
            float Dy[H], Sy[H];
#pragma unroll
            for (int h = 0; h < H; ++h) { Dy[h] = q2N[h] - q2M[h]; Sy[h] = q2N[h] + (q2M[h] + HEPSQ); }
            // ---- x faces (i-1/2) of my own levels
            float Fx[KB], u2[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int h = kk + 1;
                const float qL = dpp_l(q2P[h]);
                const float t = auP[kk] * (q2P[h] - qL) * frcp(q2P[h] + qL + EPSQ)
                              - cuvP[kk] * (Dy[h] + dpp_l(Dy[h])) * frcp(Sy[h] + dpp_l(Sy[h]))
                              - cuwP[kk] * (sP.Dz[kk] + dpp_l(sP.Dz[kk])) * frcp(sP.Sz[kk] + dpp_l(sP.Sz[kk]));
                u2[kk] = t; Fx[kk] = upw(qL, q2P[h], t);
            }
            // ---- z faces above levels k0-1 .. k0+KB-1 (the lowest one is also computed by the wave below)
            float DxA[H], SxA[H];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) { DxA[kk + 1] = sP.Dx[kk]; SxA[kk + 1] = sP.Sx[kk]; }
            { const float l0 = dpp_l(q2P[0]), l1 = dpp_l(q2P[H - 1]);
              DxA[0] = dpp_r(q2P[0]) - l0; SxA[0] = dpp_r2(q2P[0], l0) + (l0 + HEPSQ);
              DxA[H - 1] = dpp_r(q2P[H - 1]) - l1; SxA[H - 1] = dpp_r2(q2P[H - 1], l1) + (l1 + HEPSQ); }
            float Fz[KB + 1], w2[KB + 1];
#pragma unroll
            for (int hf = 0; hf <= KB; ++hf) {
                // coefficients already times dz (:383-385); zero for the top level (:214)
                float t = awP[hf] * (q2P[hf + 1] - q2P[hf]) * frcp(q2P[hf + 1] + q2P[hf] + EPSQ)
                        - cwuP[hf] * (DxA[hf] + DxA[hf + 1]) * frcp(SxA[hf] + SxA[hf + 1])
                        - cwvP[hf] * (Dy[hf] + Dy[hf + 1]) * frcp(Sy[hf] + Sy[hf + 1]);
                if (hf == 0) t *= gmul;                           // no face below the ground
                w2[hf] = t; Fz[hf] = upw(q2P[hf], q2P[hf + 1], t);
            }
            // ---- limiter, x direction.  First / last cell of a line: the lanes beyond the ring are copies of the ring column
            // (whose q2 == l), so max3 / min3 over (left, cell, right) ARE the reference's two-cell extrema there, and its
            // "qmax(n) without l(n)" as well; the ring's fin = fout = 0 is the factor rm in the denominator.
            float FxLim[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int h = kk + 1;
                if (FCT) {
                    // in / out flow from the positive and negative part of the one face a lane owns: max(0, F_W) - min(0, F_E), max(0, F_E) - min(0, F_W)
                    const float qc = q2P[h], Fp = fmaxf(0.f, Fx[kk]), Fm = fminf(0.f, Fx[kk]);
                    const float fin = Fp - dpp_r(Fm), fout = dpp_r(Fp) - Fm;
                    const float qmax = fmaxf(dpp_r(sP.m[kk]), opaque(fmaxf(dpp_l(sP.m[kk]), sP.m[kk])));
                    const float qmin = fminf(dpp_r(sP.n[kk]), opaque(fminf(dpp_l(sP.n[kk]), sP.n[kk])));
                    const float bin = (qmax - qc) * frcp(__builtin_fmaf(fin, rm, EPSF)), bout = (qc - qmin) * frcp(__builtin_fmaf(fout, rm, EPSF));
                    const float bLin = dpp_l(bin), bLout = dpp_l(bout);
                    const float s = fminf(1.0f, (u2[kk] > 0.0f) ? fminf(bin, bLout) : fminf(bLin, bout));
                    FxLim[kk] = s * Fx[kk];
                } else FxLim[kk] = Fx[kk];
                xdiv[kk] = dpp_r(FxLim[kk]) - FxLim[kk];
            }
            // ---- limiter, z direction
            float FzLim[KB + 1];
            if (FCT) {
                float bZin[H], bZout[H], Fzp[KB + 1], Fzm[KB + 1];
#pragma unroll
                for (int hf = 0; hf <= KB; ++hf) { Fzp[hf] = fmaxf(0.f, Fz[hf]); Fzm[hf] = fminf(0.f, Fz[hf]); }
                auto betaz = [&](const int kk) {
                    const int h = kk + 1;
                    const float qc = q2P[h];
                    float FzTp = Fzp[h], FzTm = Fzm[h];
                    const float mlo = (kk > 0) ? sP.m[kk - 1] : sP.mh0, nlo = (kk > 0) ? sP.n[kk - 1] : sP.nh0;
                    float mhi = (kk < KB - 1) ? sP.m[kk + 1] : sP.mh1, nhi = (kk < KB - 1) ? sP.n[kk + 1] : sP.nh1;
                    float mc = sP.m[kk], nc = sP.n[kk];
                    // (level 0: mlo / nlo are the cell's own extrema and FzB = +-0, which is the reference's first-cell form)
                    // last cell of the column: extrema without l(n) and without a cell above, fin = fout = |FzB|
                    if (EXACT ? (h == KB) : true) {
                        const bool tp = EXACT ? topwave : (h >= htop);
                        mc = tp ? qc : mc; nc = tp ? qc : nc; mhi = tp ? qc : mhi; nhi = tp ? qc : nhi; FzTp = tp ? Fzp[h - 1] : FzTp; FzTm = tp ? Fzm[h - 1] : FzTm;
                    }
                    const float qmax = max3f(mlo, mc, mhi), qmin = min3f(nlo, nc, nhi);
                    const float fin = Fzp[h - 1] - FzTm, fout = FzTp - Fzm[h - 1];
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
                bZin[0] = s_bz[par][wlo][bzlo][lane]; bZout[0] = s_bz[par][wlo][bzlo + 1][lane];
                bZin[H - 1] = s_bz[par][whi][bzhi][lane]; bZout[H - 1] = s_bz[par][whi][bzhi + 1][lane];
#pragma unroll
                for (int hf = 0; hf <= KB; hf += KB) {             // the two faces shared with a neighbouring wave (ground / column top: zero flux)
                    const float s = fminf(1.0f, (w2[hf] > 0.0f) ? fminf(bZin[hf + 1], bZout[hf]) : fminf(bZin[hf], bZout[hf + 1]));
                    FzLim[hf] = s * Fz[hf];
                }
            } else {
#pragma unroll
                for (int hf = 0; hf <= KB; ++hf) FzLim[hf] = Fz[hf];
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) zdiv[kk] = FzLim[kk + 1] - FzLim[kk];
        }

        // ---- the inputs of the NEXT step: in flight while this step finishes (y limiter, store, park) ----
        __builtin_amdgcn_sched_barrier(0);
        {
            const int o2 = (STEADY ? P + 2 : CLAMPJ(P + 2)) * sj4, o3 = (STEADY ? P + 3 : CLAMPJ(P + 3)) * sj4;
            ISSUE_LOADS_A(o2, o3)
        }
        __builtin_amdgcn_sched_barrier(0);

        ylim();

        // ================= S6: store plane M =================
        const int M = P - 1;
        if (STEADY || (M >= ja && M <= jb)) {
            if (lane_store) {
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                    if (EXACT || (kk >= kst0 && kk <= kst1)) stb(out, vk[kk + 1], M * sj4, outM[kk]);
            }
        }
        // the boundary rows of the new field are the old ones (adv_mpdata.f90:63-65)
        if (!STEADY && M == 0 && ja == 1) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
                if (lane_store && (EXACT || (kk >= kst0 && kk <= kst1))) stb(out, vk[kk + 1], 0, q2M[kk + 1]);
        }
        if (!STEADY && P == ny - 1 && jb == ny - 2) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
                if (lane_store && (EXACT || (kk >= kst0 && kk <= kst1))) stb(out, vk[kk + 1], (ny - 1) * sj4, q2P[kk + 1]);
        }
        // ---- park what the next step needs of plane P
        const float rmP = (STEADY || (P > 0 && P < ny - 1)) ? rm : 0.0f;
        float rvP[NG4 * 4];
#pragma unroll
        for (int t = 0; t < NG4; ++t) { const float4 v = s_rv[par ^ 1][t][tid]; rvP[4 * t] = v.x; rvP[4 * t + 1] = v.y; rvP[4 * t + 2] = v.z; rvP[4 * t + 3] = v.w; }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int h = kk + 1;
            // 1 / (jaco rho), 1 / (dz jaco rho) of plane P as its donor-cell pass computed them a step ago (registers / LDS), zero on the
            // ring: no load (every load in flight costs this kernel ~1 %)
            const float rdh = sP.rh[kk] * rmP, rdv = rvP[kk] * rmP;
            const float accP = q2P[h] - (xdiv[kk] - FyLimS[kk]) * rdh - zdiv[kk] * rdv;
            s_park[NA4 + 2 * kk][tid] = make_float4(sP.m[kk], sP.n[kk], v2N[kk], FyN[kk]);
            s_park[NA4 + 2 * kk + 1][tid] = make_float4(bYin[kk], bYout[kk], accP, rdh);
        }
        {
            float pq[NA4 * 4];
#pragma unroll
            for (int t = 0; t < NA4 * 4; ++t) pq[t] = (t < H) ? q2P[t < H ? t : 0] : 0.f;
#pragma unroll
            for (int t = 0; t < NA4; ++t) s_park[t][tid] = make_float4(pq[4 * t], pq[4 * t + 1], pq[4 * t + 2], pq[4 * t + 3]);
        }
        if (!STEADY) sP = sN;                                      // the generic form rolls by copying
    };
    using MdGeneric = MpMode<false, 0>; using MdEven = MpMode<true, 0>; using MdOdd = MpMode<true, 1>;
    {
        // steady steps: P in [s0, s1] (possibly empty), taken in pairs; s0 - P0 = 4, so a pair starts on an even step parity.
        // A steady step addresses planes P-1 .. P+3 without clamping: P + 3 <= ny - 1.  The last step of a chunk, P = jb + 1 (it
        // completes and stores plane jb), is a steady one too wherever those planes exist: what it computes for plane jb + 1 is not stored.
        const int s0 = ja + 1, s1 = max(min(jb + 1, ny - 4), s0 - 1);
        const int npair = (s1 - s0 + 1) / 2, s1p = s0 + 2 * npair - 1;
        for (int ph = 0; ph < 2; ++ph) {                           // generic warm-up, steady bulk, generic tail
            const int lo = ph ? s1p + 1 : P0, hi = ph ? jb + 1 : s0 - 1;
            for (int P = lo; P <= hi; ++P) step(MdGeneric{}, Q0, Q1, S0, S1, P);
            if (ph == 0)
                for (int P = s0; P < s1p; P += 2) { step(MdEven{}, Q0, Q1, S0, S1, P); step(MdOdd{}, Q1, Q0, S1, S0, P + 1); }
        }
    }
#undef ISSUE_LOADS_A
#undef LDQ
#undef LDC
#undef MP_POST
#undef MP_WAIT
#undef CLAMPJ
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int KB>
static void launch_fused(icar_hip_ctx *c, bool fct, bool pass1, bool exact, const CVarPtrs &in, const VarPtrs &out, int nv,
                         int nw, int clen, int ntile, int nchunk, int nkr, int kstore)
{
    const unsigned nitem = (unsigned)(ntile * nchunk * nkr * nv), cap = (nitem + 7u) / 8u;
    const dim3 g(8u * cap), b(64, nw);                      // block id = xcd + 8 * slot, slot < cap
#define GO(F, P1, E) hipLaunchKernelGGL((k_mpdata_fused<KB, F, P1, E>), g, b, 0, c->stream, c->d, in, out, c->U, c->V, c->W, c->mpc, (unsigned)(c->n3 * sizeof(float)), clen, ntile, nchunk, nv, nkr, kstore)
#define GO2(F, P1) { if (exact) GO(F, P1, true); else GO(F, P1, false); }
    if (fct) { if (pass1) GO2(true, true) else GO2(true, false) } else { if (pass1) GO2(false, true) else GO2(false, false) }
#undef GO2
#undef GO
}

// the scalar-independent coefficients of this step's Courant winds (c->U, c->V, c->Wdz), for icar_hip_setup_winds
int icar_mpdata_coef_run(icar_hip_ctx *c, bool rho_on)
{
    const float *jaco = icar_field_f(c, ICAR_F_JACOBIAN), *dz = icar_field_f(c, ICAR_F_ADVECTION_DZ);
    const float *rho = rho_on ? icar_field_f(c, ICAR_F_DENSITY) : nullptr;
    if (!jaco || !dz || (rho_on && !rho)) return 1;
    if (c->n3 * sizeof(float) * MPC_N >= ((size_t)1 << 32)) { icar_set_error("mpdata: a tile of more than 97 M cells is not supported (32-bit buffer offsets into the coefficient arrays)"); return 1; }
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
    const bool exact = (nkr == 1) && (kb * nw == nz);          // the waves hold exactly the column: no per-slot level tests
#define KBCASE(K) case K: launch_fused<K>(c, fct, pass1, exact, in, out, nv, nw, clen, ntile, nchunk, nkr, kstore); break;
    switch (kb) { KBCASE(1) KBCASE(2) KBCASE(3) KBCASE(4) KBCASE(5) default: icar_set_error("mpdata: internal level-range error"); return 1; }
#undef KBCASE
    HIPCHK(hipGetLastError());
    return 0;
}
