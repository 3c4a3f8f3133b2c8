/* oracle/step_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * CPU restatement of the streaming rows between the hot kernels of step():
 *   diagnostic_update   src/main/time_step.f90:49-198
 *   apply_forcing       src/objects/domain_obj.f90:2383-2448
 *   enforce_limits      src/objects/domain_obj.f90:2228-2243
 *   balance_uvw         src/physics/wind.f90:81-169 (+ calc_divergence :172-228)
 *   iterative_winds     src/physics/wind.f90:371-498 (stages split so a tiled run can exchange_u/v between them)
 *   compute_dt (3)      src/main/time_step.f90:264-289
 * PINNED by execution (tests/test_oracle_helpers_vs_ref.py, vs utilities/atm_utilities.f90 compiled unmodified into
 * oracle/_ref): exner_function, compute_ivt, compute_iq.
 * PARITY UNPINNED by execution for the rest: these procedures live in NetCDF/coarray-dependent units
 * (domain_obj.f90, time_step.f90, wind.f90) that cannot be compiled in this image; the code below is
 * restated from the source statement by statement.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#define IDX(i,k,j) ((size_t)(i) + (size_t)nx*((size_t)(k) + (size_t)nz*(size_t)(j)))
#define IDXU(i,k,j) ((size_t)(i) + (size_t)(nx+1)*((size_t)(k) + (size_t)nz*(size_t)(j)))

extern int g_math_mode;
static inline float st_powf(float x, float y) { return g_math_mode ? (float)pow((double)x, (double)y) : powf(x, y); }
static const float Rd = 287.058f, cp = 1012.0f;

void orc_diagnostic_update(int nx, int nz, int ny, const float *p, const float *th, const float *u, const float *v, const float *w,
                           const float *dzdx, const float *dzdy, const float *jaco,
                           float *exner, float *p_i, float *psfc, float *T, float *T_i, float *rho, float *u_mass, float *v_mass, float *w_real)
{
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i) {
        const size_t c = IDX(i, k, j);
        exner[c] = st_powf(p[c] / 100000.0f, Rd / cp);          /* atm_utilities.f90:682-691 */
        T[c] = th[c] * exner[c];
    }
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i) {
        const size_t c = IDX(i, k, j);
        if (k == 0) {
            p_i[c] = p[c] + (p[c] - p[IDX(i, 1, j)]) / 2;
            T_i[c] = T[c] + (T[c] - T[IDX(i, 1, j)]) / 2;
            psfc[i + (size_t)nx * j] = p_i[c];
        } else {
            p_i[c] = (p[IDX(i, k - 1, j)] + p[c]) / 2;
            T_i[c] = (T[IDX(i, k - 1, j)] + T[c]) / 2;
        }
        rho[c] = p[c] / (Rd * T[c]);
        u_mass[c] = (u[IDXU(i + 1, k, j)] + u[IDXU(i, k, j)]) / 2;
        v_mass[c] = (v[IDX(i, k, j + 1)] + v[c]) / 2;
    }
    for (int j = 1; j < ny - 1; ++j) for (int i = 1; i < nx - 1; ++i) {
        float lastw = 0;
        for (int k = 0; k < nz; ++k) {
            const size_t c = IDX(i, k, j);
            const float uw0 = u[IDXU(i, k, j)] * dzdx[IDXU(i, k, j)], uw1 = u[IDXU(i + 1, k, j)] * dzdx[IDXU(i + 1, k, j)];
            const float vw0 = v[c] * dzdy[c], vw1 = v[IDX(i, k, j + 1)] * dzdy[IDX(i, k, j + 1)];
            const float currw = w[c];
            w_real[c] = (uw0 + uw1) * 0.5f + (vw0 + vw1) * 0.5f + jaco[c] * (lastw + currw) * 0.5f;
            lastw = currw;
        }
    }
}

/* x has extents (nxm, nz, nym) (staggered fields pass their own extents) */
/* compute_ivt / compute_iq, src/utilities/atm_utilities.f90:35-64 / :73-102 (q2d accumulates over k = kms..kme-1) */
void orc_compute_ivt(int nx, int nz, int ny, const float *qv, const float *u, const float *v, const float *p_i, float *ivt)
{
    const float gravity = 9.81f;
    for (size_t t = 0; t < (size_t)nx * ny; ++t) ivt[t] = 0;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz - 1; ++k) for (int i = 0; i < nx; ++i) {
        const size_t c = IDX(i, k, j), o = (size_t)i + (size_t)nx * j;
        const float sp = sqrtf(u[c] * u[c] + v[c] * v[c]);
        if (p_i[IDX(i, k + 1, j)] > 50000) ivt[o] = ivt[o] + (qv[c] * sp * (p_i[c] - p_i[IDX(i, k + 1, j)])) / gravity;
        else if (p_i[c] > 50000) ivt[o] = ivt[o] + (qv[c] * sp * (p_i[c] - 50000)) / gravity;
    }
}

void orc_compute_iq(int nx, int nz, int ny, const float *q, const float *p_i, float *iq)
{
    const float gravity = 9.81f;
    for (size_t t = 0; t < (size_t)nx * ny; ++t) iq[t] = 0;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz - 1; ++k) for (int i = 0; i < nx; ++i) {
        const size_t c = IDX(i, k, j), o = (size_t)i + (size_t)nx * j;
        if (p_i[IDX(i, k + 1, j)] > 50000) iq[o] = iq[o] + (q[c] * (p_i[c] - p_i[IDX(i, k + 1, j)])) / gravity;
        else if (p_i[c] > 50000) iq[o] = iq[o] + (q[c] * (p_i[c] - 50000)) / gravity;
    }
}

void orc_apply_forcing(int nxm, int nz, int nym, float *x, const float *dqdt, double dt, int force_boundaries,
                       int west, int east, int south, int north)
{
    for (int j = 0; j < nym; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nxm; ++i) {
        const size_t c = (size_t)i + (size_t)nxm * ((size_t)k + (size_t)nz * j);
        int doit = 1;
        if (force_boundaries)
            doit = (west && i == 0 && j > 0 && j < nym - 1) || (east && i == nxm - 1 && j > 0 && j < nym - 1)
                || (south && j == 0) || (north && j == nym - 1);
        if (doit) x[c] = (float)((double)x[c] + ((double)dqdt[c] * dt));
    }
}

void orc_enforce_limits(size_t n, float *x) { for (size_t c = 0; c < n; ++c) if (x[c] < 0) x[c] = 0; }

void orc_balance_uvw(int nx, int nz, int ny, const float *u, const float *v, float *w, const float *ju, const float *jv,
                     const float *jw, const float *dz, float dx)
{
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
        float wprev = 0, jwprev = 0;
        for (int k = 0; k < nz; ++k) {
            const size_t c = IDX(i, k, j);
            const float du = u[IDXU(i + 1, k, j)] * ju[IDXU(i + 1, k, j)] - u[IDXU(i, k, j)] * ju[IDXU(i, k, j)];
            const float dv = v[IDX(i, k, j + 1)] * jv[IDX(i, k, j + 1)] - v[c] * jv[c];
            const float div = (du + dv) / dx;
            float wk;
            if (k == 0) wk = 0 - div * dz[c] / jw[c];
            else wk = (wprev * jwprev - div * dz[c]) / jw[c];
            w[c] = wk; wprev = wk; jwprev = jw[c];
        }
    }
}

/* make_winds_grid_relative (wind.f90:236-287), PARITY UNPINNED (wind.f90 needs FFTW3 through linear_winds; restated statement
 * by statement with Fortran's whole-array semantics: every right-hand side is evaluated before its assignment).
 * u (nx+1,nz,ny), v (nx,nz,ny+1) REAL(4) in place; sintheta, costheta REAL(8) (nx,ny).  The mixed REAL*DOUBLE products are
 * formed in double and rounded once on assignment to the REAL u_local / v_local. */
void orc_make_winds_grid_relative(int nx, int nz, int ny, float *u, float *v, const double *sintheta, const double *costheta)
{
    float *ul = (float *)malloc(sizeof(float) * (size_t)(nx + 1)), *vl = (float *)malloc(sizeof(float) * (size_t)nx);
    /* :254 u(:ime,:,:) = (u(:ime,:,:) + u(ims+1:,:,:))/2 ; :255 v(:,:,:jme) = (v(:,:,:jme) + v(:,:,jms+1:))/2 */
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i)
        u[IDXU(i, k, j)] = (u[IDXU(i, k, j)] + u[IDXU(i + 1, k, j)]) / 2;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i)
        v[IDX(i, k, j)] = (v[IDX(i, k, j)] + v[IDX(i, k, j + 1)]) / 2;
    /* :257-265 rotate */
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) {
        for (int i = 0; i < nx; ++i) {
            const double c = costheta[i + (size_t)nx * j], s = sintheta[i + (size_t)nx * j];
            ul[i] = (float)((double)u[IDXU(i, k, j)] * c - (double)v[IDX(i, k, j)] * s);
            vl[i] = (float)((double)v[IDX(i, k, j)] * c + (double)u[IDXU(i, k, j)] * s);
        }
        for (int i = 0; i < nx; ++i) { u[IDXU(i, k, j)] = ul[i]; v[IDX(i, k, j)] = vl[i]; }
    }
    /* :270-272: back onto the staggered grid; the two lost cells are extrapolated */
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) {
        for (int i = 0; i < nx; ++i) ul[i] = u[IDXU(i, k, j)];
        for (int i = 1; i < nx; ++i) u[IDXU(i, k, j)] = (ul[i - 1] + ul[i]) / 2;
        u[IDXU(0, k, j)] = 2 * u[IDXU(0, k, j)] - u[IDXU(1, k, j)];
        u[IDXU(nx, k, j)] = 2 * u[IDXU(nx - 1, k, j)] - u[IDXU(nx - 2, k, j)];
    }
    /* :274-276 */
    {
        float *col = (float *)malloc(sizeof(float) * (size_t)ny);
        for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i) {
            for (int j = 0; j < ny; ++j) col[j] = v[IDX(i, k, j)];
            for (int j = 1; j < ny; ++j) v[IDX(i, k, j)] = (col[j - 1] + col[j]) / 2;
            v[IDX(i, k, 0)] = 2 * v[IDX(i, k, 0)] - v[IDX(i, k, 1)];
            v[IDX(i, k, ny)] = 2 * v[IDX(i, k, ny - 1)] - v[IDX(i, k, ny - 2)];
        }
        free(col);
    }
    free(ul); free(vl);
}

/* calc_divergence, full form (wind.f90:203-226): horizontal + vertical metric divergence, then / jaco */
void orc_calc_divergence(int nx, int nz, int ny, const float *u, const float *v, const float *w, const float *ju, const float *jv,
                         const float *jw, const float *dz, const float *jaco, float dx, float *div)
{
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i) {
        const size_t c = IDX(i, k, j);
        const float du = u[IDXU(i + 1, k, j)] * ju[IDXU(i + 1, k, j)] - u[IDXU(i, k, j)] * ju[IDXU(i, k, j)];
        const float dv = v[IDX(i, k, j + 1)] * jv[IDX(i, k, j + 1)] - v[c] * jv[c];
        float d = (du + dv) / dx;
        const float wm = w[c] * jw[c];
        if (k == 0) d = d + wm / dz[c];
        else d = d + (wm - w[IDX(i, k - 1, j)] * jw[IDX(i, k - 1, j)]) / dz[c];
        div[c] = d / jaco[c];
    }
}

/* wind.f90:430-441: remove the model-top w linearly with the fractional height of each level (k ascending, so the
 * top level itself is zeroed last and every lower level sees the uncorrected top value) */
void orc_iterative_winds_correct_w(int nx, int nz, int ny, float *w, const float *dz)
{
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
        float height = 0;
        for (int k = 0; k < nz; ++k) height += dz[IDX(i, k, j)];
        float part = 0;
        for (int k = 0; k < nz; ++k) {
            part += dz[IDX(i, k, j)];
            float corr = part / height;
            if (corr > 1.0f) corr = 1.0f;
            w[IDX(i, k, j)] = w[IDX(i, k, j)] - corr * w[IDX(i, nz - 1, j)];
        }
    }
}

/* one pass of the loop body wind.f90:455-481 (without the exchanges): ADJ = div / (-2/dx), then the four array
 * statements on u and v with U_cor = V_cor = 0.5.  adj is nx*nz*ny scratch. */
void orc_iterative_winds_sweep(int nx, int nz, int ny, float *u, float *v, const float *w, const float *ju, const float *jv,
                               const float *jw, const float *dz, const float *jaco, float dx, float *adj)
{
    const float coef = -2 / dx;
    orc_calc_divergence(nx, nz, ny, u, v, w, ju, jv, jw, dz, jaco, dx, adj);
    for (size_t c = 0; c < (size_t)nx * nz * ny; ++c) adj[c] = adj[c] / coef;
    for (int j = 1; j <= ny - 2; ++j) for (int k = 0; k < nz; ++k) for (int i = 2; i <= nx - 1; ++i) {
        float x = u[IDXU(i, k, j)];
        x = x + (adj[IDX(i - 1, k, j)] * 0.5f);
        x = x - (adj[IDX(i, k, j)] * 0.5f);
        u[IDXU(i, k, j)] = x;
    }
    for (int j = 2; j <= ny - 1; ++j) for (int k = 0; k < nz; ++k) for (int i = 1; i <= nx - 2; ++i) {
        float x = v[IDX(i, k, j)];
        x = x + (adj[IDX(i, k, j - 1)] * 0.5f);
        x = x - (adj[IDX(i, k, j)] * 0.5f);
        v[IDX(i, k, j)] = x;
    }
}

float orc_max_courant(int nx, int nz, int ny, const float *u, const float *v, const float *w, const float *dzl, float dx)
{
    float m = 0;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) {
        const int zo = (k == 0) ? 0 : -1;
        for (int i = 0; i < nx; ++i) {
            const float cur = fmaxf(fabsf(u[IDXU(i, k, j)]), fabsf(u[IDXU(i + 1, k, j)])) / dx
                            + fmaxf(fabsf(v[IDX(i, k, j)]), fabsf(v[IDX(i, k, j + 1)])) / dx
                            + fmaxf(fabsf(w[IDX(i, k, j)]), fabsf(w[IDX(i, k + zo, j)])) / dzl[k];
            m = fmaxf(m, cur);
        }
    }
    return m;
}
