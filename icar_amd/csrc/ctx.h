// icar_amd/csrc/ctx.h -- device context behind the C ABI (include/icar_hip.h).
// Replaces the reference's module-level SAVE state (adv_upwind/adv_mpdata U_m,V_m,W_m;
// microphysics SR,last_model_time; Thompson tables) with one handle per image/GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <map>
#include <vector>
#include "../../include/icar_hip.h"

#define ICAR_MAX_ADV 12

struct Dims {
    int nx, nz, ny;      // memory extents (ime-ims+1, ...)
    int sk, sj;          // strides of k and j (sk = nx, sj = nx*nz)
    __host__ __device__ inline int idx(int i, int k, int j) const { return i + nx * (k + nz * j); }
};

struct VarPtrs { float *p[ICAR_MAX_ADV]; };
struct CVarPtrs { const float *p[ICAR_MAX_ADV]; };

struct TimingGroup { double total_ms = 0; int launches = 0; };

struct ThompsonTables;   // mp_thompson.hip
struct LinWinds;         // linear_winds.hip
struct Wsm3State;        // mp_wsm3.hip
struct Wsm6State;        // mp_wsm6.hip
struct IcarComm;         // comm.hip

// state of the step driver (timestep.hip): the options / grid members the sub-step loop reads, the model clock, and what
// mp_driver.f90 keeps in SAVE variables (last_model_time)

struct IcarStepState {
    bool configured = false;
    icar_hip_step_config cfg;
    std::vector<float> dz_levels;
    double model_time = 0.0;                 // domain%model_time%seconds()
    double mp_last_model_time = -999.0;      // mp_driver.f90:44 last_model_time
    int winds_scheme = 0, winds_dens = 0; float winds_dt = 0.f;   // what the Courant winds on the device were set up for
    float *h_val = nullptr;                  // pinned: the reduced CFL maximum
    bool early_open = false, early_wreal = false, early_face = false;   // a sub-step whose dt-independent opening is already in flight
    bool failed = false;                 // a sub-step was abandoned half-applied (timestep.hip: update_dt_opened): the fields are not a model state any more
    bool winds_first = true;                 // wind.f90:297 `.not. allocated(domain%sintheta)`: update_winds has not run yet
};

// the arrays of icar_hip_ctx::mpc, each of the tile's shape (k_mpdata_coef in mpdata.hip says what they hold): the first nine are the
// antidiffusive coefficients of the x / y / z faces, the last two the denominators jaco rho and dz jaco rho of the donor-cell passes
enum { MPC_AU = 0, MPC_CUV, MPC_CUW, MPC_AV, MPC_CVU, MPC_CVW, MPC_AW, MPC_CWU, MPC_CWV, MPC_GH, MPC_GV, MPC_N };

struct icar_hip_ctx {
    int device = 0;
    int ims, ime, kms, kme, jms, jme;
    Dims d;
    size_t n3 = 0;                       // nx*nz*ny
    hipStream_t stream = nullptr;        // the stream every entry point launches on (main, or aux between aux_begin/aux_end)
    bool own_stream = false;
    // second HIP stream (north_star: halo strips + exchange on one stream, interior work beside them on the other)
    hipStream_t aux = nullptr, main_saved = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool on_aux = false;
    void *field[ICAR_N_FIELDS] = {nullptr};
    float *dqdt[ICAR_N_FIELDS] = {nullptr};    // variable_t%dqdt_3d mirrors (apply_forcing)
    // advection scratch (A1-A5)
    float *U = nullptr, *V = nullptr, *W = nullptr, *Wdz = nullptr;
    float *mpc = nullptr;                // MPDATA: the MPC_N scalar-independent coefficient arrays of this step's winds (mpdata.hip)
    int mpc_dens = -1;
    float *alt[ICAR_N_ADVECTABLE] = {nullptr};   // ping-pong partner of each advected scalar
    float *mpx_buf = nullptr;            // mpdata_exact.hip: q2 (x2), u2, v2, w2 of mpx_nv scalars and the limited velocities of one
    int mpx_nv = 0;
    int mpdata_exact = 0;                // icar_hip_mpdata_exact: corrective iterations in the reference's operation order (bit-exact)
    bool winds_valid = false;
    // u / v / w bookkeeping for the prefetched CFL reduction (icar_hip_max_courant_prefetch): every entry point that writes a wind
    // field bumps the version; a raw device pointer to one of them handed out (icar_hip_field_device_ptr) disables the cache
    unsigned long long wind_version = 0;
    bool wind_ptr_escaped = false;
    struct { bool valid = false, reduced = false; unsigned long long ver = 0; float dx = 0.f; std::vector<float> dzl; } cfl_pre;
    float *h_cfl_pre = nullptr;          // pinned host copy of the prefetched maximum (d_red[8] on the device)
    bool cfl_wait_failed = false;        // the wait for that copy failed (compute_dt reports it; a NaN maximum otherwise means NaN winds)
    hipEvent_t cfl_ev = nullptr;
    float *iw_adj = nullptr;             // iterative_winds ADJ scratch (iterative_winds.hip)
    float *wgr_tmp = nullptr;            // make_winds_grid_relative: rotated mass-grid u | v (2 x n3)
    // reductions / flags
    float *d_red = nullptr;              // small device scratch for reductions
    std::vector<float> dzl_host;         // dz_levels last uploaded behind d_red (compute_dt re-sends them only when they change)
    int *d_flag = nullptr;
    ThompsonTables *thompson = nullptr;
    LinWinds *linwinds = nullptr;
    Wsm3State *wsm3 = nullptr;
    Wsm6State *wsm6 = nullptr;
    IcarComm *comm = nullptr;            // halo transport + co_min (comm.hip)
    IcarStepState step;                  // timestep.hip
    // timing
    bool timing = false;
    std::string timing_only;             // ",group,group," or empty = every group (icar_hip_timing_groups)
    std::vector<hipEvent_t> event_pool;  // timing events are reused, not created per scope
    std::map<std::string, TimingGroup> timers;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
};

void icar_set_error(const std::string &msg);
int icar_hip_check(hipError_t e, const char *what);
#define HIPCHK(x) do { if (icar_hip_check((x), #x)) return 1; } while (0)

// field geometry
size_t icar_field_count(const icar_hip_ctx *c, int field);
float *icar_field_f(icar_hip_ctx *c, int field, bool required = true);   // allocates lazily

// timing helpers
struct ScopedTimer {
    icar_hip_ctx *c; const char *group; hipEvent_t e0 = nullptr, e1 = nullptr;
    ScopedTimer(icar_hip_ctx *c_, const char *g);
    ~ScopedTimer();
};

// kernels implemented in the .hip files
int icar_advect_setup_winds(icar_hip_ctx *c, int scheme, float dt, float dx, int advect_density, bool with_wreal = false, bool *wreal_done = nullptr);
int icar_advect_run(icar_hip_ctx *c, int scheme, int order, int fct, int advect_density, const int *fields, int n);
int icar_mp_simple_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte, int *err);
int icar_mp_simple_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte, int *err);
int icar_mp_run(icar_hip_ctx *c, double dt_in, int halo, int subset);      // halo / subset < 0: argument not present
int icar_update_dt(icar_hip_ctx *c, double *seconds);
int icar_substep(icar_hip_ctx *c, double dt, bool enforce);
int icar_halo_pack(icar_hip_ctx *c, int dir, int halo, const int *fields, int n, float *buf, bool unpack);
int icar_halo_pack_dirs(icar_hip_ctx *c, int ndir, const int *dirs, int halo, const int *fields, int n, void *const *bufs, bool unpack);
int icar_max_courant_run(icar_hip_ctx *c, float dx, const float *dz_levels, float *out, float *d_out);
int icar_max_abs_winds_run(icar_hip_ctx *c, float *out3);
int icar_balance_uvw_run(icar_hip_ctx *c, float dx, int update);
int icar_iterative_winds_correct_w(icar_hip_ctx *c, int update);
int icar_mass_conservative_acceleration(icar_hip_ctx *c, int update);
int icar_iterative_winds_sweep(icar_hip_ctx *c, float dx, int nsweeps, int update);
int icar_make_winds_grid_relative(icar_hip_ctx *c, int update);
int icar_box_copy(icar_hip_ctx *c, int field, int which, int i0, int ni, int j0, int nj, float *buf, bool unpack);
enum { ICAR_DIAG_CELL = 4, ICAR_DIAG_FACE = 8 };     // finer parts of icar_diagnostic_update_run (step.hip), internal
int icar_diagnostic_update_run(icar_hip_ctx *c, int parts);
int icar_max_courant_prefetch_run(icar_hip_ctx *c, float dx, const float *dz_levels, bool allreduce);
bool icar_cfl_prefetched_global(icar_hip_ctx *c, float dx, const float *dz_levels, float *value);
bool icar_cfl_prefetch_waiting(icar_hip_ctx *c);      // a prefetched CFL maximum of the current winds is waiting for update_dt
inline void icar_winds_changed(icar_hip_ctx *c) { c->winds_valid = false; ++c->wind_version; }   // u, v, w (or density / jacobians) rewritten
int icar_apply_forcing_run(icar_hip_ctx *c, double dt, const int *fields, const int *fb, int n, int w, int e, int s, int nn);
int icar_enforce_limits_run(icar_hip_ctx *c, const int *fields, int n);
int icar_thompson_init_run(icar_hip_ctx *c, const float *params, const int *flags);
int icar_thompson_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte,
                      int ids, int ide, int jds, int jde, int kds, int kde);
int icar_thompson_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int (*tiles)[4], int kts, int kte,
                            int ids, int ide, int jds, int jde, int kds, int kde);
int icar_thompson_prepare_constants(icar_hip_ctx *c);
void icar_thompson_free(icar_hip_ctx *c);
void icar_linwinds_free(icar_hip_ctx *c);
void icar_wsm3_free(icar_hip_ctx *c);
int icar_wsm3_init_run(icar_hip_ctx *c);
int icar_wsm3_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte);
void icar_wsm6_free(icar_hip_ctx *c);
int icar_wsm6_init_run(icar_hip_ctx *c);
int icar_wsm6_run(icar_hip_ctx *c, float dt, int its, int ite, int jts, int jte, int kts, int kte);
int icar_wsm6_run_tiles(icar_hip_ctx *c, float dt, int ntiles, const int tiles[][4], int kts, int kte);
int icar_linwinds_setup_run(icar_hip_ctx *c, const icar_hip_lt_options *o, const float *terrain, int nxg, int nyg, int ids, int jds, float dx);
int icar_linear_perturbation_run(icar_hip_ctx *c, float U, float V, float Nsq, float zb, float zt, float minimum_step, double *u_out, double *v_out);
int icar_linwinds_build_lut_run(icar_hip_ctx *c, const float *zb, const float *zt, int nlev);
int icar_linwinds_build_lut_varying_run(icar_hip_ctx *c, const float *zb3, const float *zt3, int nlev);
int icar_linwinds_lut_copy(icar_hip_ctx *c, int comp, float *host, int to_dev);
int icar_linwinds_lut_entry(icar_hip_ctx *c, int comp, int k, int i, int j, float *host);
int icar_linwinds_pert_copy(icar_hip_ctx *c, int comp, float *host, int to_dev);
int icar_linwinds_terrain_frequency(icar_hip_ctx *c, double *out, size_t cap, int *fnx, int *fny);
int icar_spatial_winds_run(icar_hip_ctx *c, int update);
int icar_thompson_table_download(icar_hip_ctx *c, const char *name, double *out, size_t cap, size_t *n_out);
