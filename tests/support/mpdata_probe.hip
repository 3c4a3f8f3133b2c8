// tests/support/mpdata_probe.hip -- TEST INFRASTRUCTURE, NOT PRODUCT (built into tests/support/libicar_probe.so).
// The fused MPDATA kernel (icar_amd/csrc/mpdata.hip) never writes the field after its donor-cell pass (q2) to memory, and q2 has to
// be bit-identical to the reference's: next to the ring the flux limiter turns one ulp of q2 into a whole antidiffusive flux
// (adv_mpdata_FCT_core.f90:80-113 with fin = fout = 0).  With the nine antidiffusive coefficient arrays of the context zeroed every
// pseudo-velocity is zero, every corrective flux is zero, and the kernel's output IS q2: tests/test_gpu_advect.py compares that with
// the oracle's donor-cell pass bit for bit.
#include <hip/hip_runtime.h>
#include "ctx.h"

extern "C" int icar_probe_mpdata_zero_antidiffusion(void *ctx)
{
    icar_hip_ctx *c = (icar_hip_ctx *)ctx;
    if (!c || !c->mpc) return 1;
    static_assert(MPC_CWV == 8 && MPC_GH == 9, "the antidiffusive coefficients are arrays 0..8 of icar_hip_ctx::mpc");
    if (hipSetDevice(c->device) != hipSuccess) return 2;
    if (hipMemsetAsync(c->mpc, 0, c->n3 * sizeof(float) * MPC_GH, c->stream) != hipSuccess) return 2;
    return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : 2;
}
