"""oracle/ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end for oracle/_ref/libicar_ref.so: the *reference's own kernels*
(/root/reference/src/physics/{adv_mpdata,advect,mp_simple,mp_thompson}.f90) compiled
unmodified by oracle/build_ref.sh behind our bind(C) shim oracle/ref_shim.f90.

Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  All arrays are numpy float32, Fortran (i,k,j) order
expressed as C-order arrays of shape (ny, nz, nx)  [x fastest].

The reference keeps module-level wind arrays sized at first call: one grid size per
process for the advection entry points (use `run_isolated` for several sizes).
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libicar_ref.so")
_lib = None

THOMPSON_DEFAULTS = np.array(
    # Nt_c, TNO, am_s, rho_g, av_s, bv_s, fv_s, av_g, bv_g, av_i, Ef_si, Ef_rs, Ef_rg, Ef_ri,
    # C_cubes, C_sqrd, mu_r, t_adjust      (src/objects/options_obj.f90:1259-1284)
    [100.e6, 5.0, 0.069, 500.0, 40.0, 0.55, 100.0, 442.0, 0.89, 1847.5, 0.05, 0.95, 0.75, 0.95,
     0.5, 0.3, 0.0, 0.0], dtype=np.float32)


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libicar_ref.so not built (run oracle/build_ref.sh "
                               "in the container that has /root/reference)")
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(x):
    return ctypes.c_float(float(x))


def advect(scheme, q, u, v, w, rho, jaco, jaco_u, jaco_v, jaco_w, dz3d, dz_levels, dx, dt,
           advect_density=False, mpdata_order=2, fct=True, nsteps=1):
    """q: (nvars, ny, nz, nx) float32, advanced in place.  scheme 1=upwind 2=mpdata."""
    nvars, ny, nz, nx = q.shape
    assert u.shape == (ny, nz, nx + 1) and v.shape == (ny + 1, nz, nx) and w.shape == (ny, nz, nx)
    lib().ref_advect(ctypes.c_int(scheme), ctypes.c_int(nx), ctypes.c_int(nz), ctypes.c_int(ny),
                     ctypes.c_int(nvars), _p(q), _p(u), _p(v), _p(w), _p(rho), _p(jaco), _p(jaco_u),
                     _p(jaco_v), _p(jaco_w), _p(dz3d), _p(np.ascontiguousarray(dz_levels, np.float32)),
                     _f(dx), _f(dt), ctypes.c_int(int(advect_density)), ctypes.c_int(mpdata_order),
                     ctypes.c_int(int(fct)), ctypes.c_int(nsteps))
    return q


def mp_simple(pressure, th, pii, rho, qv, qc, qr, qs, rain, snow, dt, dz, its, ite, jts, jte, kts, kte):
    """All 3-D arrays (ny,nz,nx) float32 in place; rain/snow (ny,nx). 1-based inclusive tile bounds."""
    ny, nz, nx = qv.shape
    lib().ref_mp_simple(ctypes.c_int(nx), ctypes.c_int(nz), ctypes.c_int(ny), _p(pressure), _p(th),
                        _p(pii), _p(rho), _p(qv), _p(qc), _p(qr), _p(qs), _p(rain), _p(snow), _f(dt),
                        _p(dz), *[ctypes.c_int(int(x)) for x in (its, ite, jts, jte, kts, kte)])


def thompson_init(params=None, flags=(0, 0), workdir=None):
    """Runs the reference thompson_init; it caches its tables as *.dat in the CWD (56 s cold)."""
    params = THOMPSON_DEFAULTS if params is None else np.asarray(params, np.float32)
    fl = np.asarray(flags, np.int32)
    cwd = os.getcwd()
    if workdir:
        os.makedirs(workdir, exist_ok=True)
        os.chdir(workdir)
    try:
        lib().ref_thompson_init(_p(np.ascontiguousarray(params)), fl.ctypes.data_as(ctypes.c_void_p))
    finally:
        os.chdir(cwd)


def thompson(qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, dt, rainnc, rainncv, snownc, graupelnc, sr,
             ids, ide, jds, jde, kds, kde, its, ite, jts, jte, kts, kte):
    ny, nz, nx = qv.shape
    lib().ref_thompson(ctypes.c_int(nx), ctypes.c_int(nz), ctypes.c_int(ny),
                       _p(qv), _p(qc), _p(qr), _p(qi), _p(qs), _p(qg), _p(ni), _p(nr), _p(th), _p(pii),
                       _p(p), _p(dz), _f(dt), _p(rainnc), _p(rainncv), _p(snownc), _p(graupelnc), _p(sr),
                       *[ctypes.c_int(int(x)) for x in (ids, ide, jds, jde, kds, kde,
                                                        its, ite, jts, jte, kts, kte)])


THOMPSON_TABLES = ["tcg_racg", "tmr_racg", "tcr_gacr", "tmg_gacr", "tnr_racg", "tnr_gacr", "tcs_racs1", "tmr_racs1", "tcs_racs2",
                   "tmr_racs2", "tcr_sacr1", "tms_sacr1", "tcr_sacr2", "tms_sacr2", "tnr_racs1", "tnr_racs2", "tnr_sacr1", "tnr_sacr2",
                   "tpi_qcfz", "tni_qcfz", "tpi_qrfz", "tpg_qrfz", "tni_qrfz", "tnr_qrfz", "tps_iaus", "tni_iaus", "tpi_ide",
                   "t_Efrw", "t_Efsw", "rate_constants", "size_bins"]


def thompson_table(name):
    """One of the reference's module-level Thompson tables (after thompson_init) as a flat float64 array."""
    tid = THOMPSON_TABLES.index(name) + 1
    buf = np.empty(28 * 28 * 37 * 37, np.float64)
    fn = lib().ref_thompson_table
    fn.restype = ctypes.c_int
    m = fn(ctypes.c_int(tid), ctypes.c_int(buf.size), buf.ctypes.data_as(ctypes.c_void_p))
    if m < 0:
        raise KeyError(name)
    return buf[:m].copy()


GRID_MEMBERS = ["yimg", "ximg", "yimages", "ximages", "ims", "ime", "jms", "jme", "kms", "kme", "ns_halo_nx", "ew_halo_ny",
                "halo_nz", "halo_size", "nx_global", "ny_global", "nx", "ny", "nz", "ids", "ide", "jds", "jde", "kds", "kde",
                "its", "ite", "jts", "jte", "kts", "kte", "is2d", "is3d"]


def grid(nx, ny, nz, nimages, image, nx_extra=0, ny_extra=0):
    """The reference's grid_t%set_grid_dimensions(..., for_image=image) in an nimages-image run
    (src/objects/grid_obj.f90:140-255) -> dict of its integer members."""
    out = (ctypes.c_int * 33)()
    lib().ref_grid(ctypes.c_int(nx), ctypes.c_int(ny), ctypes.c_int(nz), ctypes.c_int(nimages), ctypes.c_int(image),
                   ctypes.c_int(nx_extra), ctypes.c_int(ny_extra), out)
    return dict(zip(GRID_MEMBERS, list(out)))


# ---- helper modules compiled unmodified: utilities/atm_utilities.f90, utilities/array_utilities.f90 -------------------
def _i(x):
    return ctypes.c_int(int(x))


def exner(p):
    p = np.ascontiguousarray(p, np.float32); out = np.empty_like(p)
    lib().ref_exner(_i(p.size), _p(p), _p(out))
    return out


def wind_polar(u, v):
    """calc_direction, calc_speed and calc_u / calc_v of those: (direction, speed, u_back, v_back)."""
    u = np.ascontiguousarray(u, np.float32); v = np.ascontiguousarray(v, np.float32)
    o = [np.empty_like(u) for _ in range(4)]
    lib().ref_wind_polar(_i(u.size), _p(u), _p(v), *[_p(x) for x in o])
    return o


def calc_stability(th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc):
    a = [np.ascontiguousarray(x, np.float32) for x in (th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc)]
    out = np.empty_like(a[0])
    lib().ref_calc_stability(_i(a[0].size), *[_p(x) for x in a], _p(out))
    return out


def compute_ivt(qv, u, v, p_i):
    ny, nz, nx = qv.shape
    out = np.zeros((ny, nx), np.float32)
    lib().ref_compute_ivt(_i(nx), _i(nz), _i(ny), _p(qv), _p(u), _p(v), _p(p_i), _p(out))
    return out


def compute_iq(q, p_i):
    ny, nz, nx = q.shape
    out = np.zeros((ny, nx), np.float32)
    lib().ref_compute_iq(_i(nx), _i(nz), _i(ny), _p(q), _p(p_i), _p(out))
    return out


def linear_space(vmin, vmax, n):
    out = np.zeros(n, np.float32)
    lib().ref_linear_space(_i(n), _f(vmin), _f(vmax), _p(out))
    return out


def calc_weight(axis, bestpos, match):
    axis = np.ascontiguousarray(axis, np.float32); bestpos = np.ascontiguousarray(bestpos, np.int32)
    match = np.ascontiguousarray(match, np.float32)
    nextpos = np.zeros(bestpos.size, np.int32); w = np.zeros(bestpos.size, np.float32)
    lib().ref_calc_weight(_i(axis.size), _p(axis), _i(bestpos.size), bestpos.ctypes.data_as(ctypes.c_void_p), _p(match),
                          nextpos.ctypes.data_as(ctypes.c_void_p), _p(w))
    return nextpos, w


def smooth_array_3d(a, windowsize, ydim=3):
    """a: numpy (ny, nz, nx) == Fortran (nx, nz, ny); smoothed in place."""
    ny, nz, nx = a.shape
    lib().ref_smooth_array_3d(_i(nx), _i(nz), _i(ny), _p(a), _i(windowsize), _i(ydim))
    return a


# ---- WSM3 (physics/mp_wsm3.f90 compiled unmodified) ------------------------------------------------------------------
def wsm3_init():
    """wsm3init as mp_driver.f90:105 calls it -> (module constants in the order of oracle.orc.WSM3_CONSTS, the 18 scalar
    arguments mp_driver.f90:554-585 passes to wsm3; element 0 = delt is left 0)."""
    out = np.zeros(44, np.float32); args = np.zeros(18, np.float32)
    lib().ref_wsm3_init(_p(out), _p(args))
    return out[:42].copy(), args


def wsm3(th, q, qci, qrs, w, den, pii, p, delz, delt, rain, rainncv, snow, snowncv, sr, its, ite, jts, jte, kts, kte):
    ny, nz, nx = q.shape
    lib().ref_wsm3(_i(nx), _i(nz), _i(ny), _p(th), _p(q), _p(qci), _p(qrs), _p(w), _p(den), _p(pii), _p(p), _p(delz), _f(delt),
                   _p(rain), _p(rainncv), _p(snow), _p(snowncv), _p(sr), *[_i(x) for x in (its, ite, jts, jte, kts, kte)])


# ---- WSM6 (physics/mp_wsm6.f90 compiled unmodified) ------------------------------------------------------------------
def wsm6_init():
    """wsm6init as mp_driver.f90:100 calls it -> the 60 module constants in the order of oracle.orc.WSM6_CONSTS"""
    out = np.zeros(60, np.float32)
    lib().ref_wsm6_init(_p(out))
    return out


def wsm6(th, q, qc, qr, qi, qs, qg, den, pii, p, delz, delt, rain, rainncv, sr, snow, graupel, its, ite, jts, jte, kts, kte):
    ny, nz, nx = q.shape
    lib().ref_wsm6(_i(nx), _i(nz), _i(ny), _p(th), _p(q), _p(qc), _p(qr), _p(qi), _p(qs), _p(qg), _p(den), _p(pii), _p(p), _p(delz),
                   _f(delt), _p(rain), _p(rainncv), _p(sr), _p(snow), _p(graupel), *[_i(x) for x in (its, ite, jts, jte, kts, kte)])
