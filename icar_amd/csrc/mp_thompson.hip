// icar_amd/csrc/mp_thompson.hip -- Thompson microphysics (rows M2-M4): placeholder until the
// column kernel lands; the entry points fail loudly rather than fall back to anything.
#include "ctx.h"
int icar_thompson_init_run(icar_hip_ctx *, const float *, const int *) { icar_set_error("thompson_init: not implemented in this build"); return 1; }
int icar_thompson_run(icar_hip_ctx *, float, int, int, int, int, int, int, int, int, int, int, int, int) { icar_set_error("thompson: not implemented in this build"); return 1; }
void icar_thompson_free(icar_hip_ctx *) {}
