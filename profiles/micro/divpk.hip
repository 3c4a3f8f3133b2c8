// profiles/micro/divpk.hip -- measurement aid (not product): two IEEE FP32 divisions with the six fma/mul of the
// compiler's expansion issued as packed v_pk_*_f32.  Checks bit-equality with `/` on random + edge operands and times both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 fdiv2(f2 n, f2 d)
{
    bool vcc0, vcc1, dummy;
    f2 ds, ns;
    ds.x = __builtin_amdgcn_div_scalef(n.x, d.x, false, &dummy);     // scaled denominator
    ns.x = __builtin_amdgcn_div_scalef(n.x, d.x, true, &vcc0);       // scaled numerator + flag
    ds.y = __builtin_amdgcn_div_scalef(n.y, d.y, false, &dummy);
    ns.y = __builtin_amdgcn_div_scalef(n.y, d.y, true, &vcc1);
    f2 r; r.x = __builtin_amdgcn_rcpf(ds.x); r.y = __builtin_amdgcn_rcpf(ds.y);
    const f2 one = {1.0f, 1.0f};
    f2 e = __builtin_elementwise_fma(-ds, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    f2 q = ns * r;
    f2 rem = __builtin_elementwise_fma(-ds, q, ns);
    q = __builtin_elementwise_fma(rem, r, q);
    rem = __builtin_elementwise_fma(-ds, q, ns);
    f2 o;
    o.x = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(rem.x, r.x, q.x, vcc0), d.x, n.x);
    o.y = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(rem.y, r.y, q.y, vcc1), d.y, n.y);
    return o;
}

__global__ void k_ref(const float *n, const float *d, float *o, int m)
{
    int i = blockIdx.x * 256 + threadIdx.x; if (2 * i + 1 >= m) return;
    o[2 * i] = n[2 * i] / d[2 * i]; o[2 * i + 1] = n[2 * i + 1] / d[2 * i + 1];
}
__global__ void k_pk(const float *n, const float *d, float *o, int m)
{
    int i = blockIdx.x * 256 + threadIdx.x; if (2 * i + 1 >= m) return;
    f2 a = {n[2 * i], n[2 * i + 1]}, b = {d[2 * i], d[2 * i + 1]};
    f2 r = fdiv2(a, b); o[2 * i] = r.x; o[2 * i + 1] = r.y;
}
template <int PK> __global__ void k_chain(const float *n, const float *d, float *o, int reps)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    f2 a = {n[2 * i], n[2 * i + 1]}, b = {d[2 * i], d[2 * i + 1]}, acc = {0, 0};
    for (int r = 0; r < reps; ++r) {
        f2 q;
        if (PK) q = fdiv2(a, b); else { q.x = a.x / b.x; q.y = a.y / b.y; }
        acc += q; a += acc * 1e-3f; b += 1e-3f;
    }
    o[2 * i] = acc.x; o[2 * i + 1] = acc.y;
}
int main()
{
    const int m = 1 << 22;
    std::vector<float> n(m), d(m), r0(m), r1(m);
    srand(1);
    auto rnd = [] { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); float f; memcpy(&f, &u, 4); return f; };
    for (int i = 0; i < m; ++i) { n[i] = rnd(); d[i] = rnd(); }
    const float edge[] = {0.f, -0.f, 1.f, -1.f, 1e-45f, 1e-39f, 1e38f, 3e38f, INFINITY, -INFINITY, NAN, 1e-20f, 1e20f, 0.333333f};
    int e = 0; for (float a : edge) for (float b : edge) { n[e] = a; d[e] = b; ++e; }
    for (int i = 1000; i < m / 2; ++i) { n[i] = (float)(rand() % 100000) * 1e-9f; d[i] = 1.0f + (rand() % 1000) * 1e-3f; }   // scheme-like magnitudes
    float *dn, *dd, *o0, *o1; hipMalloc(&dn, m * 4); hipMalloc(&dd, m * 4); hipMalloc(&o0, m * 4); hipMalloc(&o1, m * 4);
    hipMemcpy(dn, n.data(), m * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), m * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ref, dim3(m / 512), dim3(256), 0, 0, dn, dd, o0, m);
    hipLaunchKernelGGL(k_pk, dim3(m / 512), dim3(256), 0, 0, dn, dd, o1, m);
    hipMemcpy(r0.data(), o0, m * 4, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), o1, m * 4, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < m; ++i) { bool nn = r0[i] != r0[i] && r1[i] != r1[i]; if (!nn && memcmp(&r0[i], &r1[i], 4)) { if (bad < 5) printf("diff %g / %g : %g vs %g\n", n[i], d[i], r0[i], r1[i]); ++bad; } }
    printf("bit differences: %ld of %d\n", bad, m);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k_chain<1>, dim3(m / 512), dim3(256), 0, 0, dn, dd, o1, 200);
        else hipLaunchKernelGGL(k_chain<0>, dim3(m / 512), dim3(256), 0, 0, dn, dd, o0, 200);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms\n", pk ? "packed pair" : "compiler /  ", ms);
    }
    return 0;
}
