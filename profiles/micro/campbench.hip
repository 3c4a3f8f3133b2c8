// profiles/micro/campbench.hip -- do N arrays read in lockstep at the same offset collide on the HBM channels when their bases are
// 40 MiB apart (the size of a 512 x 40 x 512 REAL(4) field)?  Reads N arrays of 40 MiB, writes one; bases 40 MiB apart against
// bases staggered by k * stagger bytes.   hipcc --offload-arch=gfx950 -O3 campbench.hip -o campbench && ./campbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N>
__global__ void __launch_bounds__(256) k_sum(const float *base, size_t stride, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < N; ++a) s += base[a * stride + i];
    out[i] = s;
}
template <int N>
static double run(const float *base, size_t stride, float *out, size_t n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned g = (unsigned)((n + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_sum<N>, dim3(g), dim3(256), 0, 0, base, stride, out, n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_sum<N>, dim3(g), dim3(256), 0, 0, base, stride, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)(N + 1) * n * 4 * 10 / (ms * 1e-3) / 1e9;
}
int main()
{
    const size_t n = (size_t)512 * 40 * 512;                 // 40 MiB per array
    const size_t maxpad = 1 << 20;                           // floats
    float *buf, *out;
    hipMalloc(&buf, (16 * (n + maxpad)) * sizeof(float)); hipMalloc(&out, n * sizeof(float));
    hipMemset(buf, 0, (16 * (n + maxpad)) * sizeof(float));
    const size_t pads[] = {0, 64, 1024, 1024 + 64, 4096 + 256, 65536 + 1024 + 64, 262144 + 4096 + 64};   // floats added to the 40 MiB stride
    for (size_t pad : pads) {
        printf("stride 40 MiB + %8zu B :  2 arrays %7.0f   4 arrays %7.0f   8 arrays %7.0f   12 arrays %7.0f   14 arrays %7.0f GB/s\n", pad * 4,
               run<2>(buf, n + pad, out, n), run<4>(buf, n + pad, out, n), run<8>(buf, n + pad, out, n), run<12>(buf, n + pad, out, n), run<14>(buf, n + pad, out, n));
    }
    return 0;
}
