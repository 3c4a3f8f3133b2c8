// tests/support/th_probe.hip -- TEST INFRASTRUCTURE, NOT PRODUCT: evaluates the device functions of the Thompson level code
// (icar_amd/csrc/thompson_math.h: the very header mp_thompson.hip compiles) on arrays of arguments, so that the parity tests can
// compare them with the host's libm / with the reference's index loop value by value.  Built by tests/support/build_probe.py into
// tests/support/libicar_probe.so; libicar_hip.so exports no probe.
#include <hip/hip_runtime.h>
#include "thompson_math.h"

namespace {
__global__ void k_p10(float *p10) { for (int n = 0; n < TH_P10_N; ++n) p10[n] = powi10f(n - TH_P10_OFF); }     // as k_thompson_constants fills ThState::p10

// the decade index of the level code for n values: which = 0 the product's form (fast path + fallback), 1 the reference's loop alone
__global__ void k_dec_index(const float *__restrict__ p10, const float *__restrict__ rf, const double *__restrict__ rd, int n, int n2, int which, int *__restrict__ out)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const DK K_ = d_consts();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (rf) out[t] = which ? dec_index_f_slow(K_, rf[t], n2) : dec_index_f_k(K_, p10, rf[t], n2);
    else    out[t] = which ? dec_index_d_slow(K_, rd[t], n2) : dec_index_d_k(K_, p10, rd[t], n2);
}
// DOUBLE PRECISION sites: op 0 d_log(x), 1 d_exp(x), 2 d_pow(x, y).  REAL(4) sites (arguments narrowed, results widened): 3 powf(x, y),
// 4 expf(x), 5 logf(x), 6 log10f(x), 7 atanf(x), 8 powf through the shared-base form (d_powf_base + d_powf_l), 9 10.**x.
__global__ void k_math(int op, int n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out)
{
    th_lds_init(threadIdx.x, blockDim.x);
    const DK K_ = d_consts();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double r;
    if (op == 0) r = d_log(x[t]);
    else if (op == 1) r = d_exp(x[t]);
    else if (op == 2) r = d_pow(x[t], y[t]);
    else if (op == 3) r = (double)d_powf((float)x[t], (float)y[t]);
    else if (op == 4) r = (double)d_expf((float)x[t]);
    else if (op == 5) r = (double)gf_logf((float)x[t]);
    else if (op == 6) r = (double)d_log10f((float)x[t]);
    else if (op == 7) r = (double)gf_atanf((float)x[t]);
    else if (op == 8) { const PowBase b = d_powf_base((float)x[t]); r = (double)d_powf_l(b, (float)y[t]); }
    else r = (double)d_pow10f((float)x[t]);
    out[t] = r;
}
#define CK(x) do { if ((x) != hipSuccess) return 2; } while (0)
}  // namespace

extern "C" {
// host arrays of n doubles; y may be NULL unless op is 2, 3 or 8.  Returns 0, 1 (bad arguments) or 2 (HIP error).
int icar_probe_math(int op, int n, const double *x, const double *y, double *out)
{
    if (op < 0 || op > 9 || !x || !out || ((op == 2 || op == 3 || op == 8) && !y)) return 1;
    if (n <= 0) return 0;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    CK(hipMalloc(&dx, sizeof(double) * n)); CK(hipMalloc(&dout, sizeof(double) * n));
    CK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    if (y) { CK(hipMalloc(&dy, sizeof(double) * n)); CK(hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, dx, dy, dout);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout); if (dy) (void)hipFree(dy);
    return 0;
}
// exactly one of r4 / r8; n2 = the table's first decade
int icar_probe_dec_index(const float *r4, const double *r8, int n, int n2, int which, int *out)
{
    if (!out || (!r4 && !r8) || (r4 && r8)) return 1;
    if (n <= 0) return 0;
    float *drf = nullptr, *p10 = nullptr; double *drd = nullptr; int *dout = nullptr;
    CK(hipMalloc(&dout, sizeof(int) * n)); CK(hipMalloc(&p10, sizeof(float) * TH_P10_N));
    hipLaunchKernelGGL(k_p10, dim3(1), dim3(1), 0, 0, p10);
    if (r4) { CK(hipMalloc(&drf, sizeof(float) * n)); CK(hipMemcpy(drf, r4, sizeof(float) * n, hipMemcpyHostToDevice)); }
    else    { CK(hipMalloc(&drd, sizeof(double) * n)); CK(hipMemcpy(drd, r8, sizeof(double) * n, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_dec_index, dim3((n + 255) / 256), dim3(256), 0, 0, p10, drf, drd, n, n2, which ? 1 : 0, dout);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, sizeof(int) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dout); (void)hipFree(p10); if (drf) (void)hipFree(drf); if (drd) (void)hipFree(drd);
    return 0;
}
}
