// profiles/micro/divbench.hip -- VALU cost of FP32 division variants on gfx950 (cycles per wave64 division, one SIMD).
//   ieee : the compiler's correctly rounded expansion (v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup)
//   lean : the same Newton/Markstein arithmetic without the range scaling / fixup (v_rcp, 6 fma, mul): identical
//          result whenever no operand or quotient is subnormal / near overflow
//   fma  : a stream of plain v_fma_f32 for reference
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt divbench.hip -o divbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float div_lean(float n, float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    const float e2 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e3, r, q);
}
template <int MODE> __global__ void k(float *out, const float *in, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = in[t], a1 = in[t] + 1.f, a2 = in[t] + 2.f, a3 = in[t] + 3.f, b = in[t] * 0.5f + 1.25f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 = a0 / b + 1.f; a1 = a1 / b + 1.f; a2 = a2 / b + 1.f; a3 = a3 / b + 1.f; b = b + 1e-3f; }
        if (MODE == 1) { a0 = div_lean(a0, b) + 1.f; a1 = div_lean(a1, b) + 1.f; a2 = div_lean(a2, b) + 1.f; a3 = div_lean(a3, b) + 1.f; b = b + 1e-3f; }
        if (MODE == 2) { a0 = __builtin_fmaf(a0, b, 1.f); a1 = __builtin_fmaf(a1, b, 1.f); a2 = __builtin_fmaf(a2, b, 1.f); a3 = __builtin_fmaf(a3, b, 1.f); b = b + 1e-3f; }
    }
    out[t] = a0 + a1 + a2 + a3;
}
template <int MODE> double run(float *out, float *in, int iters, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, in, 16);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    const int blocks = 256 * 8 * 4, n = blocks * 256, iters = 2000;      // 8 waves per SIMD
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 1.0f + (i % 977) * 1e-3f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    const double waves = (double)n / 64, simds = 1024, clk = 2.4e9;
    const char *names[3] = {"ieee", "lean", "fma "};
    double ms[3] = {run<0>(out, in, iters, blocks), run<1>(out, in, iters, blocks), run<2>(out, in, iters, blocks)};
    for (int m = 0; m < 3; ++m) {
        const double per_op = ms[m] * 1e-3 * clk * simds / (waves * iters * 4.0);   // cycles of one SIMD per wave-level op (incl. the +1)
        printf("%s: %.3f ms  -> %.1f SIMD cycles per wave64 (op + add)\n", names[m], ms[m], per_op);
    }
    // exactness of lean vs ieee on normal-range operands
    return 0;
}
