/* oracle/wind_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * CPU restatement of SURVEY.md section 8 row W2, statement by statement, loops in the reference's order:
 *   spatial_winds        src/physics/linear_winds.f90:840-1127   (reverse=.false.)
 *   calc_stability       src/utilities/atm_utilities.f90:401-467 (dry / moist / sat lapse rate)
 *   calc_direction/speed src/utilities/atm_utilities.f90:334-367
 *   smooth_array (ydim=3) src/utilities/array_utilities.f90:308-417
 *   calc_weight          src/utilities/array_utilities.f90:263-288
 * PINNED by execution (tests/test_oracle_helpers_vs_ref.py, bit-exact vs atm_utilities.f90 / array_utilities.f90
 * compiled unmodified into oracle/_ref): calc_stability, calc_direction, calc_speed, calc_weight, smooth_array_3d.
 * PARITY UNPINNED by execution: the body of spatial_winds itself -- linear_winds.f90 needs FFTW3 + the NetCDF/coarray
 * domain object and cannot be compiled in this image; no reference test holds expected values for it
 * (test_caf_linear_winds_setup.f90 is a smoke test).  The LUT-build half (row W3, FFT) is restated in oracle/wind_oracle.py.
 *
 * Index convention: arrays are Fortran order, 0-based here.  X(i,k,j) -> i + nx*(k + nz*j).
 * Math mode (icar_oracle.c: orc_set_math_mode): 0 = libm logf/expf/atanf as the compiled Fortran calls them,
 * 1 = FP64 function rounded once (what the HIP kernels evaluate).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

extern int g_math_mode;
static inline float w_logf(float x) { return g_math_mode ? (float)log((double)x) : logf(x); }
static inline float w_expf(float x) { return g_math_mode ? (float)exp((double)x) : expf(x); }
static inline float w_atanf(float x) { return g_math_mode ? (float)atan((double)x) : atanf(x); }

static const float pi = 3.1415927f;                 /* icar_constants.f90:395 */
static const float LH_vaporization = 2260000.0f, Rd = 287.058f, Rw = 461.5f, cp = 1012.0f, gravity = 9.81f;

typedef struct {                                    /* options%lt_options members spatial_winds reads */
    int variable_N, smooth_nsq;
    float N_squared, max_stability, min_stability, linear_contribution, linear_update_fraction;
    int n_dir, n_spd, n_nsq;
    const float *dir_values, *spd_values, *nsq_values;
} orc_lt_opts;

static float calc_sat_lapse_rate(float T, float mr)            /* atm_utilities.f90:401-410 */
{
    const float L = LH_vaporization;
    return gravity * ((1 + (L * mr) / (Rd * T)) / (cp + (L * L * mr * (Rd / Rw)) / (Rd * T * T)));
}

static float calc_moist_stability(float t_top, float t_bot, float z_top, float z_bot, float qv_top, float qv_bot, float qc)
{                                                               /* atm_utilities.f90:417-430 */
    const float t = (t_top + t_bot) / 2, qv = (qv_top + qv_bot) / 2, dz = z_top - z_bot;
    const float sat_lapse = calc_sat_lapse_rate(t, qv);
    return (gravity / t) * ((t_top - t_bot) / dz + sat_lapse) * (1 + (LH_vaporization * qv) / (Rd * t))
           - (gravity / (1 + qv + qc) * (qv_top - qv_bot) / dz);
}

static float calc_stability(const orc_lt_opts *o, float th_top, float th_bot, float pii_top, float pii_bot,
                            float z_top, float z_bot, float qv_top, float qv_bot, float qc)
{                                                               /* atm_utilities.f90:448-467 */
    if (qc < 1e-7f) {
        if (o->variable_N) return gravity * (w_logf(th_top) - w_logf(th_bot)) / (z_top - z_bot);   /* :436-442 */
        return o->N_squared;
    }
    if (o->variable_N) return calc_moist_stability(th_top * pii_top, th_bot * pii_bot, z_top, z_bot, qv_top, qv_bot, qc);
    return o->N_squared / 10.0f;
}

float orc_calc_direction(float u, float v)                      /* atm_utilities.f90:334-355 */
{
    if (v < 0) return w_atanf(u / v) + pi;
    if (v == 0) return (u > 0) ? pi / 2.0f : pi * 1.5f;
    if (u >= 0) return w_atanf(u / v);
    return w_atanf(u / v) + (2 * pi);
}

static float calc_weight(const float *d, int n, int bestpos, int *nextpos, float match)   /* 1-based positions */
{
    if (match < d[0]) { *nextpos = 1; return 1; }
    if (bestpos == n) { *nextpos = n; return 1; }
    *nextpos = bestpos + 1;
    return (d[*nextpos - 1] - match) / (d[*nextpos - 1] - d[bestpos - 1]);
}

/* element-wise entry points for pinning the helpers against the compiled reference (oracle/ref.py) */
void orc_calc_stability_n(int n, int variable_N, float N_squared, const float *th_top, const float *th_bot, const float *pii_top,
                          const float *pii_bot, const float *z_top, const float *z_bot, const float *qv_top, const float *qv_bot,
                          const float *qc, float *out)
{
    orc_lt_opts o = {0}; o.variable_N = variable_N; o.N_squared = N_squared;
    for (int t = 0; t < n; ++t)
        out[t] = calc_stability(&o, th_top[t], th_bot[t], pii_top[t], pii_bot[t], z_top[t], z_bot[t], qv_top[t], qv_bot[t], qc[t]);
}

void orc_calc_weight_n(int n, const float *axis, int m, const int *bestpos, const float *match, int *nextpos, float *weight)
{
    for (int t = 0; t < m; ++t) weight[t] = calc_weight(axis, n, bestpos[t], &nextpos[t], match[t]);
}

/* smooth_array_3d(wind, windowsize, ydim=3): wind is (nx, nlev, nrow) */
void orc_smooth_array_ydim3(int nx, int nlev, int nrow, float *wind, int w)
{
    const size_t n = (size_t)nx * nlev * nrow;
    float *in = (float *)malloc(n * sizeof(float));
    double *rowsums = (double *)malloc(nx * sizeof(double)), *rowmeans = (double *)malloc(nx * sizeof(double));
    for (size_t t = 0; t < n; ++t) in[t] = wind[t];
    const int nrows = w * 2 + 1, ncols = w * 2 + 1;
#define IN(i, j, k) in[(size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)nlev * (size_t)(k))]
    for (int j = 0; j < nlev; ++j) {
        for (int i = 0; i < nx; ++i) rowsums[i] = (double)(IN(i, j, 0) * (float)(w + 2));
        const int lim = w < nrow ? w : nrow;
        for (int r = 2; r <= lim; ++r) for (int i = 0; i < nx; ++i) rowsums[i] = rowsums[i] + IN(i, j, r - 1);
        if (w > nrow) for (int i = 0; i < nx; ++i) rowsums[i] = rowsums[i] + (double)(IN(i, j, nrow - 1) * (float)(w - nrow));
        for (int k = 1; k <= nrow; ++k) {
            const int starty = (k - w > 2) ? k - w : 2, endy = (k + w < nrow) ? k + w : nrow;
            for (int i = 0; i < nx; ++i) {
                rowsums[i] = rowsums[i] - IN(i, j, starty - 2) + IN(i, j, endy - 1);
                rowmeans[i] = rowsums[i] / nrows;
            }
            double cursum = 0;                                  /* sum(rowmeans(2:w)) */
            for (int i = 2; i <= w; ++i) cursum += rowmeans[i - 1];
            cursum = cursum + rowmeans[0] * (w + 2);
            for (int i = 1; i <= nx; ++i) {
                const int startx = (i - w > 2) ? i - w : 2, endx = (i + w < nx) ? i + w : nx;
                cursum = cursum - rowmeans[startx - 2] + rowmeans[endx - 1];
                wind[(size_t)(i - 1) + (size_t)nx * ((size_t)j + (size_t)nlev * (size_t)(k - 1))] = (float)(cursum / ncols);
            }
        }
    }
#undef IN
    free(in); free(rowsums); free(rowmeans);
}

/* spatial_winds(domain, reverse=.false., vsmooth, winsz, update): u3d/v3d are either data_3d or dqdt_3d.
 * u3d (nx+1,nz,ny), v3d (nx,nz,ny+1), all other 3-D fields (nx,nz,ny).  qc/qi/qr/qs may be NULL (not associated).
 * u_lut (n_spd,n_dir,n_nsq,nx+1,nz,ny), v_lut (n_spd,n_dir,n_nsq,nx,nz,ny+1); u_pert/v_pert = hi_[uv]_perturbation. */
void orc_spatial_winds(int nx, int nz, int ny, float *u3d, float *v3d, float *nsquared,
                       const float *th, const float *exner, const float *z, const float *qv,
                       const float *qc, const float *qi, const float *qr, const float *qs,
                       const float *u_lut, const float *v_lut, float *u_pert, float *v_pert,
                       const orc_lt_opts *o, int vsmooth, int winsz)
{
    const int nxu = nx + 1, nyv = ny + 1;
#define C3(i, k, j) ((size_t)(i) + (size_t)nx * ((size_t)(k) + (size_t)nz * (size_t)(j)))
#define U3(i, k, j) ((size_t)(i) + (size_t)nxu * ((size_t)(k) + (size_t)nz * (size_t)(j)))
    /* ---- N^2 per cell, log, vertical smoothing :906-975 (1-based loop variables as in the source) */
    for (int k = 1; k <= ny; ++k) {
        for (int j = 1; j <= nz; ++j) {
            for (int i = 1; i <= nx; ++i) {
                float val;
                if (o->variable_N) {
                    const int top = (j + vsmooth < nz) ? j + vsmooth : nz;
                    const int b0 = j - (vsmooth - (top - j));
                    const int bottom = b0 > 1 ? b0 : 1;
                    float hydrometeors = 0;
                    const size_t c = C3(i - 1, j - 1, k - 1);
                    if (qc) hydrometeors = hydrometeors + qc[c];
                    if (qi) hydrometeors = hydrometeors + qi[c];
                    if (qr) hydrometeors = hydrometeors + qr[c];
                    if (qs) hydrometeors = hydrometeors + qs[c];
                    const size_t cb = C3(i - 1, bottom - 1, k - 1), ct = C3(i - 1, top - 1, k - 1);
                    /* the call passes the bottom values in the *_top slots and vice versa (:933-939) */
                    val = calc_stability(o, th[cb], th[ct], exner[cb], exner[ct], z[cb], z[ct], qv[cb], qv[ct], hydrometeors);
                    val = fmaxf(o->min_stability, fminf(o->max_stability, val));
                } else {
                    val = o->N_squared;
                }
                nsquared[C3(i - 1, j - 1, k - 1)] = val;
            }
            for (int i = 0; i < nx; ++i) nsquared[C3(i, j - 1, k - 1)] = w_logf(nsquared[C3(i, j - 1, k - 1)]);
        }
        if (o->smooth_nsq) {
            for (int j = 1; j <= nz; ++j) {
                const int top = (j + vsmooth < nz) ? j + vsmooth : nz;
                const int b0 = j - (vsmooth - (top - j));
                const int bottom = b0 > 1 ? b0 : 1;
                for (int s = bottom; s <= j - 1; ++s)
                    for (int i = 0; i < nx; ++i) nsquared[C3(i, j - 1, k - 1)] = nsquared[C3(i, j - 1, k - 1)] + nsquared[C3(i, s - 1, k - 1)];
                for (int s = j + 1; s <= top; ++s)
                    for (int i = 0; i < nx; ++i) nsquared[C3(i, j - 1, k - 1)] = nsquared[C3(i, j - 1, k - 1)] + nsquared[C3(i, s - 1, k - 1)];
                for (int i = 0; i < nx; ++i) nsquared[C3(i, j - 1, k - 1)] = nsquared[C3(i, j - 1, k - 1)] / (float)(top - bottom + 1);
            }
        }
    }
    if (o->smooth_nsq) orc_smooth_array_ydim3(nx, nz, ny, nsquared, winsz);

    /* ---- LUT interpolation :990-1122 */
    float *u1d = (float *)malloc(nxu * sizeof(float)), *v1d = (float *)malloc(nxu * sizeof(float));
    const int ns = o->n_spd, nd = o->n_dir, nn = o->n_nsq;
#define ULUT(s, d, n, i, j, k) u_lut[(size_t)((s) - 1) + (size_t)ns * ((size_t)((d) - 1) + (size_t)nd * ((size_t)((n) - 1) + (size_t)nn * ((size_t)((i) - 1) + (size_t)nxu * ((size_t)((j) - 1) + (size_t)nz * (size_t)((k) - 1)))))]
#define VLUT(s, d, n, i, j, k) v_lut[(size_t)((s) - 1) + (size_t)ns * ((size_t)((d) - 1) + (size_t)nd * ((size_t)((n) - 1) + (size_t)nn * ((size_t)((i) - 1) + (size_t)nx * ((size_t)((j) - 1) + (size_t)nz * (size_t)((k) - 1)))))]
    const float luf = o->linear_update_fraction, lc = o->linear_contribution;
    for (int k = 1; k <= nyv; ++k) {
        int uk = k < ny ? k : ny;
        for (int i = 1; i <= nxu; ++i) {
            const int vi = i < nx ? i : nx;
            float su = 0, sv = 0;
            for (int j = 0; j < nz; ++j) su = su + u3d[U3(i - 1, j, uk - 1)];
            for (int j = 0; j < nz; ++j) sv = sv + v3d[C3(vi - 1, j, k - 1)];
            u1d[i - 1] = su / (float)nz;
            v1d[i - 1] = sv / (float)nz;
        }
        for (int j = 1; j <= nz; ++j) {
            for (int i = 1; i <= nxu; ++i) {
                uk = k < ny ? k : ny;
                const int vi = i < nx ? i : nx;
                const int bottom = (j - winsz > 1) ? j - winsz : 1, top = (j + winsz < nz) ? j + winsz : nz;
                const float u = u1d[i - 1], v = v1d[i - 1];
                int dpos = 1, spos = 1, npos = 1, nextd, nexts, nextn;
                const float curdir = orc_calc_direction(u, v);
                for (int s = 1; s <= nd; ++s) if (curdir > o->dir_values[s - 1]) dpos = s;
                const float curspd = sqrtf(u * u + v * v);
                for (int s = 1; s <= ns; ++s) if (curspd > o->spd_values[s - 1]) spos = s;
                float sn = 0;
                for (int s = bottom; s <= top; ++s) sn = sn + nsquared[C3(vi - 1, s - 1, uk - 1)];
                const float curnsq = sn / (float)(top - bottom + 1);
                for (int s = 1; s <= nn; ++s) if (curnsq > o->nsq_values[s - 1]) npos = s;
                const float dweight = calc_weight(o->dir_values, nd, dpos, &nextd, curdir);
                const float sweight = calc_weight(o->spd_values, ns, spos, &nexts, curspd);
                const float nweight = calc_weight(o->nsq_values, nn, npos, &nextn, curnsq);
                if (k <= ny) {
                    const float wind_first = nweight * (dweight * ULUT(spos, dpos, npos, i, j, k) + (1 - dweight) * ULUT(spos, nextd, npos, i, j, k))
                                           + (1 - nweight) * (dweight * ULUT(spos, dpos, nextn, i, j, k) + (1 - dweight) * ULUT(spos, nextd, nextn, i, j, k));
                    const float wind_second = nweight * (dweight * ULUT(nexts, dpos, npos, i, j, k) + (1 - dweight) * ULUT(nexts, nextd, npos, i, j, k))
                                            + (1 - nweight) * (dweight * ULUT(nexts, dpos, nextn, i, j, k) + (1 - dweight) * ULUT(nexts, nextd, nextn, i, j, k));
                    const size_t c = U3(i - 1, j - 1, k - 1);
                    u_pert[c] = u_pert[c] * (1 - luf) + luf * (sweight * wind_first + (1 - sweight) * wind_second);
                    u3d[c] = u3d[c] + u_pert[c] * lc;
                }
                if (i <= nx) {
                    const float wind_first = nweight * (dweight * VLUT(spos, dpos, npos, i, j, k) + (1 - dweight) * VLUT(spos, nextd, npos, i, j, k))
                                           + (1 - nweight) * (dweight * VLUT(spos, dpos, nextn, i, j, k) + (1 - dweight) * VLUT(spos, nextd, nextn, i, j, k));
                    const float wind_second = nweight * (dweight * VLUT(nexts, dpos, npos, i, j, k) + (1 - dweight) * VLUT(nexts, nextd, npos, i, j, k))
                                            + (1 - nweight) * (dweight * VLUT(nexts, dpos, nextn, i, j, k) + (1 - dweight) * VLUT(nexts, nextd, nextn, i, j, k));
                    const size_t c = C3(i - 1, j - 1, k - 1);
                    v_pert[c] = v_pert[c] * (1 - luf) + luf * (sweight * wind_first + (1 - sweight) * wind_second);
                    v3d[c] = v3d[c] + v_pert[c] * lc;
                }
            }
        }
    }
    free(u1d); free(v1d);
    const size_t n3 = (size_t)nx * nz * ny;
    for (size_t t = 0; t < n3; ++t) nsquared[t] = w_expf(nsquared[t]);       /* :1126 */
#undef C3
#undef U3
#undef ULUT
#undef VLUT
}
