// profiles/membench.hip -- measurement aid (not product): what the memory system gives the MPDATA access
// patterns on MI355X.  (a) 1 read + 3 writes per scalar, 9 scalars looped inside the thread (pattern of
// k_mpdata_fluxes), (b) the same with the scalar on blockIdx.z, (c) 5 reads + 1 write (k_mpdata_final).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct P { float *p[12]; };
__global__ void k_a(int n, P in, P o1, P o2, P o3, int nv) {
    const int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return;
    for (int m = 0; m < nv; ++m) { float v = in.p[m][c]; o1.p[m][c] = v; o2.p[m][c] = v + 1; o3.p[m][c] = v + 2; }
}
__global__ void k_b(int n, P in, P o1, P o2, P o3) {
    const int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return; const int m = blockIdx.y;
    float v = in.p[m][c]; o1.p[m][c] = v; o2.p[m][c] = v + 1; o3.p[m][c] = v + 2;
}
__global__ void k_c(int n, P a, P b, P c1, P d, P e, P o) {
    const int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return; const int m = blockIdx.y;
    o.p[m][c] = a.p[m][c] + b.p[m][c] + c1.p[m][c] + d.p[m][c] + e.p[m][c];
}
int main() {
    const int n = 512 * 512 * 40, nv = 9;
    std::vector<P> ps(7);
    for (auto &p : ps) for (int m = 0; m < nv; ++m) { hipMalloc(&p.p[m], n * 4); hipMemset(p.p[m], 0, n * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_a, dim3((n + 255) / 256), dim3(256), 0, 0, n, ps[0], ps[1], ps[2], ps[3], nv);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("a: 1R+3W x9 in-thread loop : %.3f ms  %.2f TB/s\n", ms / 10, 16.0 * n * nv / (ms / 10 * 1e-3) / 1e12);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_b, dim3((n + 255) / 256, nv), dim3(256), 0, 0, n, ps[0], ps[1], ps[2], ps[3]);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("b: 1R+3W scalar on grid.y   : %.3f ms  %.2f TB/s\n", ms / 10, 16.0 * n * nv / (ms / 10 * 1e-3) / 1e12);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_c, dim3((n + 255) / 256, nv), dim3(256), 0, 0, n, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5]);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("c: 5R+1W scalar on grid.y   : %.3f ms  %.2f TB/s\n", ms / 10, 24.0 * n * nv / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
