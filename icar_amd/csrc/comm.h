// icar_amd/csrc/comm.h -- internal interface of comm.hip (halo transport + co_min) for the step driver (timestep.hip).
#pragma once
struct icar_hip_ctx;
struct IcarComm;
int icar_comm_halo_send(icar_hip_ctx *c, int halo, const int *fields, int nfields);
int icar_comm_halo_retrieve(icar_hip_ctx *c, int halo, const int *fields, int nfields);
int icar_comm_exchange_uv(icar_hip_ctx *c, int halo, int which);      // exchange_u + exchange_v (which = 0 data_3d, 1 dqdt_3d)
bool icar_comm_has_peers(icar_hip_ctx *c);                             // a neighbouring image on at least one side
int icar_comm_co_reduce(icar_hip_ctx *c, double *value, bool take_min);
int icar_comm_max_device(icar_hip_ctx *c, float *d_val);     // 0 done on the device, 2 = no device-side transport, 1 error
void icar_comm_free(icar_hip_ctx *c);
