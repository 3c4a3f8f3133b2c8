// profiles/micro/issuebench.hip -- measurement aid (not product): what does a NON-VALU instruction cost a wave that is
// otherwise issuing VALU work, at 1..4 waves per SIMD on gfx950?  The MPDATA steady loop carries 556 SALU / branch
// instructions beside 1103 VALU per step at 2 waves per SIMD; valubench shows one wave alone issues a VALU instruction only
// every ~6 clocks.  If SALU instructions take a wave's issue slot the same way, the scalar bookkeeping of the loop is a
// third of its time.
// build: hipcc --offload-arch=gfx950 -O3 issuebench.hip -o issuebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Mix { V_ADD, V_ADD_SADD_2_1, V_ADD_SADD_1_1, V_ADD_BR_NT, V_ADD_BR_T, V_ADD_NOP, V_FMA, V_FMA_SADD_1_1, V_ADD_EXECBR, V_ADD_SMOV64, V_ADD_WAITCNT, SADD_ONLY, V_ADD_VMOV_1_1, V_ADD_DPP_1_1, NMIX };
static const char *names[NMIX] = {"32 v_add", "32 v_add + 16 s_add", "32 v_add + 32 s_add", "32 v_add + 8 (s_cmp + cbranch not taken)", "32 v_add + 8 s_branch taken",
                                  "32 v_add + 8 s_nop 1", "32 v_fma", "32 v_fma + 32 s_add", "32 v_add + 8 (saveexec + cbranch_execz + or exec)", "32 v_add + 16 s_mov_b64",
                                  "32 v_add + 8 s_waitcnt (nothing pending)", "32 s_add", "32 v_add + 32 v_mov", "32 v_add + 32 v_mov_dpp"};

template <int MIX> __global__ void __launch_bounds__(256) k(float *out, const float *in, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], m[8];
    const float c = in[t] * 1e-3f, one = 1.0f + c * 1e-4f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = in[t] + r; m[r] = c + r; }
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MIX == V_FMA || MIX == V_FMA_SADD_1_1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(one), "v"(c));
                else if (MIX != SADD_ONLY) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
                if (MIX == V_ADD_SADD_1_1 || MIX == V_FMA_SADD_1_1 || MIX == SADD_ONLY || (MIX == V_ADD_SADD_2_1 && (r & 1))) {
                    if (r & 2) asm volatile("s_add_i32 %0, %0, %1" : "+s"(s1) : "s"(s3) : "scc"); else asm volatile("s_add_i32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc");
                }
                if (MIX == V_ADD_SMOV64 && (r & 1)) asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
                if (MIX == V_ADD_BR_NT && (r & 3) == 3) asm volatile("s_cmp_lt_i32 %0, 0\n\ts_cbranch_scc1 1f\n1:" ::"s"(s0) : "scc");
                if (MIX == V_ADD_BR_T && (r & 3) == 3) asm volatile("s_branch 1f\n\ts_nop 0\n1:" ::);
                if (MIX == V_ADD_NOP && (r & 3) == 3) asm volatile("s_nop 1");
                if (MIX == V_ADD_WAITCNT && (r & 3) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
                if (MIX == V_ADD_EXECBR && (r & 3) == 3) asm volatile("s_and_saveexec_b64 vcc, exec\n\ts_cbranch_execz 1f\n1:\n\ts_or_b64 exec, exec, vcc" ::: "vcc", "scc");
                if (MIX == V_ADD_VMOV_1_1) asm volatile("v_mov_b32 %0, %1" : "+v"(m[r]) : "v"(c));
                if (MIX == V_ADD_DPP_1_1) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(m[r]));
            }
        }
    }
    float s = (float)(s1 + s2);
#pragma unroll
    for (int r = 0; r < 8; ++r) s += a[r] + m[r];
    out[t] = s;
}

template <int MIX> static void bench(float *out, float *in)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    printf("%-52s", names[MIX]);
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = 256 * wps;                     // 256-thread blocks = one wave per SIMD each
        hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(256), 0, 0, out, in, 16);
        hipDeviceSynchronize();
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // clocks of ONE wave per iteration of the 32-VALU body (wall clocks: all waves of a SIMD run concurrently)
        printf("  %dw: %7.1f", wps, best * 1e-3 * 2.4e9 / iters);
    }
    printf("   (clocks @2.4 GHz per 32-op body, per SIMD wall)\n"); fflush(stdout);
}

int main()
{
    const int n = 256 * 4 * 256;
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 1.0f + (i % 977) * 1e-3f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    bench<V_ADD>(out, in); bench<V_ADD_SADD_2_1>(out, in); bench<V_ADD_SADD_1_1>(out, in); bench<SADD_ONLY>(out, in); bench<V_ADD_SMOV64>(out, in);
    bench<V_ADD_BR_NT>(out, in); bench<V_ADD_BR_T>(out, in); bench<V_ADD_EXECBR>(out, in); bench<V_ADD_NOP>(out, in); bench<V_ADD_WAITCNT>(out, in);
    bench<V_FMA>(out, in); bench<V_FMA_SADD_1_1>(out, in); bench<V_ADD_VMOV_1_1>(out, in); bench<V_ADD_DPP_1_1>(out, in);
    return 0;
}
