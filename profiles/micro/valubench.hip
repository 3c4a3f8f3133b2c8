// profiles/micro/valubench.hip -- issue cost of the VALU instructions the advection / microphysics kernels are made of, on
// gfx950: SIMD cycles per wave64 instruction at 1, 2, 4, 8 waves per SIMD, 8 independent chains per wave.
// Answers (round 2): is v_pk_*_f32 two results per issue slot or one?  what do v_rcp_f32 and the IEEE division pieces
// cost next to a plain v_fma_f32?  what do the cross-lane moves (DPP, ds_bpermute) cost?
// build: hipcc --offload-arch=gfx950 -O3 valubench.hip -o valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA, MUL, ADD, MINF, CNDMASK, PKFMA, PKMUL, PKADD, RCP, DIVSCALE, DIVFMAS, DIVFIXUP, IEEEDIV, RCPMUL, RCPNR, DPPMOV, BPERM, FMA64, SQRT, EXP, MOV, PKMOV, MAXF, CND_SET, CND_SGPR, FMAC, MULB, SUB, MAX3, MED3, NOPS };
static const char *names[NOPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_cndmask_b32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                  "v_rcp_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "a/b (IEEE expansion)", "a*rcp(b)", "rcp+1 Newton+mul",
                                  "v_mov_dpp row_shr:1", "ds_bpermute_b32", "v_fma_f64", "v_sqrt_f32", "v_exp_f32", "v_mov_b32", "v_pk_mov_b32", "v_max_f32", "cmp+cndmask+add (C)", "v_cndmask sgpr", "v_fmac_f32", "v_mul_f32 (b~1)", "v_sub_f32", "v_max3_f32", "v_med3_f32"};

template <int OP> __global__ void __launch_bounds__(256) k(float *out, const float *in, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8]; float2v p[8]; double dd[8];
    const float b = in[t] * 0.5f + 1.25f, c = in[t] * 1e-3f;
    const float2v pb = {b, b}, pc = {c, c};
    const double db = b, dc = c;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = in[t] + r; p[r] = float2v{a[r], a[r] + 0.5f}; dd[r] = a[r]; }
    const int idx = ((threadIdx.x + 1) & 63) * 4;
    const unsigned long long vccv = __ballot(in[t] > 1.3f);
    const float one = 1.0f + c * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(one), "v"(c));
                if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
                if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
                if (OP == MINF) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b) : );
                if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(pb), "v"(pc));
                if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(pb));
                if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(pc));
                if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[r]));
                if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[r]));
                if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[r]));
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[r]) : "v"(b));
                if (OP == PKMOV) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[r]) : "v"(pb));
                if (OP == DIVSCALE) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[r]) : "v"(b) : "vcc");
                if (OP == DIVFMAS) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c) : );
                if (OP == DIVFIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
                if (OP == IEEEDIV) a[r] = a[r] / b + 1.0f;
                if (OP == RCPMUL) a[r] = a[r] * __builtin_amdgcn_rcpf(b + a[r]);
                if (OP == RCPNR) { const float d = b + a[r]; float rr = __builtin_amdgcn_rcpf(d); rr = __builtin_fmaf(__builtin_fmaf(-d, rr, 1.0f), rr, rr); a[r] = a[r] * rr; }
                if (OP == DPPMOV) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[r]));
                if (OP == BPERM) a[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a[r])));
                if (OP == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
                if (OP == CND_SET) a[r] = (a[r] > b) ? c : a[r] + one;   // compiler's own v_cmp + v_cndmask (+ add)
                if (OP == CND_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "s"(vccv));
                if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[r]) : "v"(one), "v"(c));
                if (OP == MULB) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[r]) : "v"(one));
                if (OP == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
                if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
                if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
                if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[r]) : "v"(db), "v"(dc));
            }
        }
    }
    float s = 0; double sd = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) { s += a[r] + p[r].x + p[r].y; sd += dd[r]; }
    out[t] = s + (float)sd;
}

template <int OP> static double run1(float *out, float *in, int iters, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, in, 16);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

template <int OP> static void bench(float *out, float *in)
{
    const int iters = 500;
    printf("%-24s", names[OP]);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;                     // 256-thread blocks = 4 waves = one per SIMD; wps blocks per CU
        const double ms = run1<OP>(out, in, iters, blocks);
        const double waves_per_simd = wps, clk = 2.4e9;
        const double ops = (double)iters * 32;            // per wave
        const double cyc = ms * 1e-3 * clk / (waves_per_simd * ops);
        printf("  %dw: %6.2f", wps, cyc);
    }
    printf("   (SIMD cycles @2.4GHz per wave64 op)\n");
}

int main()
{
    const int n = 256 * 8 * 256;
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 1.0f + (i % 977) * 1e-3f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    bench<FMA>(out, in); bench<MUL>(out, in); bench<ADD>(out, in); bench<MINF>(out, in); bench<CNDMASK>(out, in); bench<MOV>(out, in);
    bench<PKFMA>(out, in); bench<PKMUL>(out, in); bench<PKADD>(out, in); bench<PKMOV>(out, in);
    bench<RCP>(out, in); bench<SQRT>(out, in); bench<EXP>(out, in);
    bench<DIVSCALE>(out, in); bench<DIVFMAS>(out, in); bench<DIVFIXUP>(out, in);
    bench<IEEEDIV>(out, in); bench<RCPMUL>(out, in); bench<RCPNR>(out, in);
    bench<MAXF>(out, in); bench<CND_SET>(out, in); bench<CND_SGPR>(out, in); bench<FMAC>(out, in); bench<MULB>(out, in); bench<SUB>(out, in); bench<MAX3>(out, in); bench<MED3>(out, in);
    bench<DPPMOV>(out, in); bench<BPERM>(out, in); bench<FMA64>(out, in);
    return 0;
}
