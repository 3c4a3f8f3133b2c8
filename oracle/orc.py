"""oracle/orc.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end for oracle/liboracle.so (the plain-C CPU restatement, oracle/icar_oracle.c).
Same call shapes as oracle/ref.py so tests can swap one for the other.
Arrays: numpy float32, C-order (ny, nz, nx) == Fortran (i,k,j).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.dtype in (np.float32, np.float64, np.int32) and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(x):
    return ctypes.c_float(float(x))


def _i(x):
    return ctypes.c_int(int(x))


def setup_winds(scheme, u, v, w, rho, jaco_u, jaco_v, jaco_w, dx, dt, advect_density=False):
    ny, nz, nx = w.shape
    U = np.zeros((ny, nz, nx), np.float32); V = np.zeros_like(U); W = np.zeros_like(U)
    lib().orc_setup_winds(_i(scheme), _i(nx), _i(nz), _i(ny), _p(u), _p(v), _p(w), _p(rho), _p(jaco_u),
                          _p(jaco_v), _p(jaco_w), _f(dx), _f(dt), _i(advect_density), _p(U), _p(V), _p(W))
    return U, V, W


def advect(scheme, q, u, v, w, rho, jaco, jaco_u, jaco_v, jaco_w, dz3d, dz_levels, dx, dt,
           advect_density=False, mpdata_order=2, fct=True, nsteps=1):
    nvars, ny, nz, nx = q.shape
    lib().orc_advect(_i(scheme), _i(nx), _i(nz), _i(ny), _i(nvars), _p(q), _p(u), _p(v), _p(w), _p(rho),
                     _p(jaco), _p(jaco_u), _p(jaco_v), _p(jaco_w), _p(dz3d), _f(dx), _f(dt),
                     _i(advect_density), _i(mpdata_order), _i(fct), _i(nsteps))
    return q


def mp_simple(pressure, th, pii, rho, qv, qc, qr, qs, rain, snow, dt, dz, its, ite, jts, jte, kts, kte):
    ny, nz, nx = qv.shape
    fn = lib().orc_mp_simple
    fn.restype = ctypes.c_int
    return fn(_i(nx), _i(nz), _i(ny), _p(pressure), _p(th), _p(pii), _p(rho), _p(qv), _p(qc), _p(qr),
              _p(qs), _p(rain), _p(snow), _f(dt), _p(dz), _i(its), _i(ite), _i(jts), _i(jte), _i(kts), _i(kte))


def set_math_mode(mode):
    """0: libm float functions (bit-identical to the compiled reference, and what the HIP kernels evaluate: glibc_flt32.h);
    1: the FP64 function rounded once (a sensitivity probe, tests/test_oracle_modes.py)."""
    lib().orc_set_math_mode(_i(mode))


def libm_f(op, x, y=None):
    """the host C library's float functions on arrays: op 3 / 8 powf(x, y), 4 expf, 5 logf, 6 log10f, 7 atanf, 9 powf(10, x)"""
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    lib().orc_libm_f(_i(op), _i(x.size), _p(x), _p(yy), _p(out))
    return out


def libm_d(op, x, y=None):
    """the host C library's double functions on arrays: op 0 log, 1 exp, 2 pow(x, y)"""
    x = np.ascontiguousarray(x, np.float64); out = np.empty_like(x)
    yy = x if y is None else np.ascontiguousarray(y, np.float64)
    lib().orc_libm_d(_i(op), _i(x.size), x.ctypes.data_as(ctypes.c_void_p), yy.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(_i(n))


def thompson_init(params, flags=(0, 0), build_tables=True):
    p = np.ascontiguousarray(params, np.float32); f = np.ascontiguousarray(flags, np.int32)
    lib().orc_thompson_init(_p(p), _p(f), _i(build_tables))


def thompson_table(name):
    fn = lib().orc_thompson_table
    fn.restype = ctypes.POINTER(ctypes.c_double)
    n = ctypes.c_size_t()
    ptr = fn(name.encode(), ctypes.byref(n))
    if not ptr:
        raise KeyError(name)
    return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()


def thompson(qv, qc, qr, qi, qs, qg, ni, nr, th, pii, p, dz, dt, rainnc, rainncv, snownc, graupelnc, sr,
             ids, ide, jds, jde, kds, kde, its, ite, jts, jte, kts, kte):
    ny, nz, nx = qv.shape
    lib().orc_thompson(_i(nx), _i(nz), _i(ny), _p(qv), _p(qc), _p(qr), _p(qi), _p(qs), _p(qg), _p(ni), _p(nr), _p(th),
                       _p(pii), _p(p), _p(dz), _f(dt), _p(rainnc), _p(rainncv), _p(snownc), _p(graupelnc), _p(sr),
                       *[_i(x) for x in (ids, ide, jds, jde, kds, kde, its, ite, jts, jte, kts, kte)])


def diagnostic_update(p, th, u, v, w, dzdx, dzdy, jaco):
    ny, nz, nx = p.shape
    out = {k: np.zeros((ny, nz, nx), np.float32) for k in ("exner", "pressure_interface", "temperature", "temperature_interface",
                                                          "density", "u_mass", "v_mass", "w_real")}
    out["surface_pressure"] = np.zeros((ny, nx), np.float32)
    lib().orc_diagnostic_update(_i(nx), _i(nz), _i(ny), _p(p), _p(th), _p(u), _p(v), _p(w), _p(dzdx), _p(dzdy), _p(jaco),
                                _p(out["exner"]), _p(out["pressure_interface"]), _p(out["surface_pressure"]), _p(out["temperature"]),
                                _p(out["temperature_interface"]), _p(out["density"]), _p(out["u_mass"]), _p(out["v_mass"]), _p(out["w_real"]))
    return out


def compute_ivt(qv, u_mass, v_mass, p_i):
    ny, nz, nx = qv.shape
    out = np.zeros((ny, nx), np.float32)
    lib().orc_compute_ivt(_i(nx), _i(nz), _i(ny), _p(qv), _p(u_mass), _p(v_mass), _p(p_i), _p(out))
    return out


def compute_iq(q, p_i):
    ny, nz, nx = q.shape
    out = np.zeros((ny, nx), np.float32)
    lib().orc_compute_iq(_i(nx), _i(nz), _i(ny), _p(q), _p(p_i), _p(out))
    return out


def calc_stability(th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc, variable_N=True, N_squared=1e-5):
    a = [np.ascontiguousarray(x, np.float32) for x in (th_top, th_bot, pii_top, pii_bot, z_top, z_bot, qv_top, qv_bot, qc)]
    out = np.zeros(a[0].size, np.float32)
    lib().orc_calc_stability_n(_i(a[0].size), _i(int(variable_N)), _f(N_squared), *[_p(x) for x in a], _p(out))
    return out


def calc_weight(axis, bestpos, match):
    axis = np.ascontiguousarray(axis, np.float32); bestpos = np.ascontiguousarray(bestpos, np.int32)
    match = np.ascontiguousarray(match, np.float32)
    nextpos = np.zeros(bestpos.size, np.int32); w = np.zeros(bestpos.size, np.float32)
    lib().orc_calc_weight_n(_i(axis.size), _p(axis), _i(bestpos.size), bestpos.ctypes.data_as(ctypes.c_void_p), _p(match),
                            nextpos.ctypes.data_as(ctypes.c_void_p), _p(w))
    return nextpos, w


def apply_forcing(x, dqdt, dt, force_boundaries, west, east, south, north):
    nym, nz, nxm = x.shape
    lib().orc_apply_forcing(_i(nxm), _i(nz), _i(nym), _p(x), _p(dqdt), ctypes.c_double(dt), _i(force_boundaries),
                            _i(west), _i(east), _i(south), _i(north))


def enforce_limits(x):
    lib().orc_enforce_limits(ctypes.c_size_t(x.size), _p(x))


def balance_uvw(u, v, ju, jv, jw, dz, dx):
    ny, nz, nx = jw.shape
    w = np.zeros((ny, nz, nx), np.float32)
    lib().orc_balance_uvw(_i(nx), _i(nz), _i(ny), _p(u), _p(v), _p(w), _p(ju), _p(jv), _p(jw), _p(dz), _f(dx))
    return w


def make_winds_grid_relative(u, v, sintheta, costheta):
    """wind.f90:236-287 in place on u (ny,nz,nx+1), v (ny+1,nz,nx); sintheta / costheta float64 (ny,nx)."""
    nyp, nz, nx = v.shape
    st = np.ascontiguousarray(sintheta, np.float64); ct = np.ascontiguousarray(costheta, np.float64)
    lib().orc_make_winds_grid_relative(_i(nx), _i(nz), _i(nyp - 1), _p(u), _p(v), st.ctypes.data_as(ctypes.c_void_p), ct.ctypes.data_as(ctypes.c_void_p))


def calc_divergence(u, v, w, ju, jv, jw, dz, jaco, dx):
    ny, nz, nx = jw.shape
    div = np.zeros((ny, nz, nx), np.float32)
    lib().orc_calc_divergence(_i(nx), _i(nz), _i(ny), _p(u), _p(v), _p(w), _p(ju), _p(jv), _p(jw), _p(dz), _p(jaco), _f(dx), _p(div))
    return div


def iterative_winds_correct_w(w, dz):
    ny, nz, nx = w.shape
    lib().orc_iterative_winds_correct_w(_i(nx), _i(nz), _i(ny), _p(w), _p(dz))


def iterative_winds_sweep(u, v, w, ju, jv, jw, dz, jaco, dx):
    ny, nz, nx = w.shape
    adj = np.zeros((ny, nz, nx), np.float32)
    lib().orc_iterative_winds_sweep(_i(nx), _i(nz), _i(ny), _p(u), _p(v), _p(w), _p(ju), _p(jv), _p(jw), _p(dz), _p(jaco), _f(dx), _p(adj))


def iterative_winds(u, v, ju, jv, jw, dz, jaco, dx, iterations):
    """wind.f90:371-498 on one image (exchange_u/v are no-ops): returns (u, v, w)."""
    u = u.copy(); v = v.copy()
    w = balance_uvw(u, v, ju, jv, jw, dz, dx)
    iterative_winds_correct_w(w, dz)
    for _ in range(iterations + 1):
        iterative_winds_sweep(u, v, w, ju, jv, jw, dz, jaco, dx)
    return u, v, w


def max_courant(u, v, w, dz_levels, dx):
    ny, nz, nx = w.shape
    fn = lib().orc_max_courant
    fn.restype = ctypes.c_float
    return float(fn(_i(nx), _i(nz), _i(ny), _p(u), _p(v), _p(w), _p(np.ascontiguousarray(dz_levels, np.float32)), _f(dx)))


# ---- W2: spatial_winds (oracle/wind_oracle.c) ---------------------------------------------------
class _lt_opts(ctypes.Structure):
    _fields_ = [("variable_N", ctypes.c_int), ("smooth_nsq", ctypes.c_int), ("N_squared", ctypes.c_float),
                ("max_stability", ctypes.c_float), ("min_stability", ctypes.c_float),
                ("linear_contribution", ctypes.c_float), ("linear_update_fraction", ctypes.c_float),
                ("n_dir", ctypes.c_int), ("n_spd", ctypes.c_int), ("n_nsq", ctypes.c_int),
                ("dir_values", ctypes.c_void_p), ("spd_values", ctypes.c_void_p), ("nsq_values", ctypes.c_void_p)]


def spatial_winds(u3d, v3d, th, exner, z, qv, hydrometeors, u_lut, v_lut, u_pert, v_pert, opt, dirv, spdv, nsqv, vsmooth, winsz):
    """orc_spatial_winds.  u3d (ny,nz,nx+1), v3d (ny+1,nz,nx) are updated in place (data_3d or dqdt_3d);
    hydrometeors = (qc, qi, qr, qs) with None for "not associated"; LUTs in the reference's order, i.e. numpy
    C-order [ny(+1), nz, nx(+1), n_nsq, n_dir, n_spd]; opt: dict.  Returns nsquared."""
    ny, nz, nx = th.shape
    nsq = np.zeros((ny, nz, nx), np.float32)
    o = _lt_opts(int(opt["variable_N"]), int(opt["smooth_nsq"]), opt["N_squared"], opt["max_stability"], opt["min_stability"],
                 opt["linear_contribution"], opt["linear_update_fraction"], len(dirv), len(spdv), len(nsqv),
                 dirv.ctypes.data, spdv.ctypes.data, nsqv.ctypes.data)
    qc, qi, qr, qs = hydrometeors
    lib().orc_spatial_winds(_i(nx), _i(nz), _i(ny), _p(u3d), _p(v3d), _p(nsq), _p(th), _p(exner), _p(z), _p(qv),
                            _p(qc), _p(qi), _p(qr), _p(qs), _p(u_lut), _p(v_lut), _p(u_pert), _p(v_pert),
                            ctypes.byref(o), _i(vsmooth), _i(winsz))
    return nsq


def smooth_array_ydim3(a, w):
    ny, nz, nx = a.shape
    lib().orc_smooth_array_ydim3(_i(nx), _i(nz), _i(ny), _p(a), _i(w))
    return a


def calc_direction(u, v):
    fn = lib().orc_calc_direction
    fn.restype = ctypes.c_float
    return float(fn(_f(u), _f(v)))


# ---- WSM3 (oracle/wsm3_oracle.c) -------------------------------------------------------------------------------------
WSM3_CONSTS = ["qc0", "qck1", "pidnc", "bvtr1", "bvtr2", "bvtr3", "bvtr4", "g1pbr", "g3pbr", "g4pbr", "g5pbro2", "pvtr", "eacrr", "pacrr",
               "precr1", "precr2", "xmmax", "roqimax", "bvts1", "bvts2", "bvts3", "bvts4", "g1pbs", "g3pbs", "g4pbs", "g5pbso2", "pvts",
               "pacrs", "precs1", "precs2", "pidn0r", "pidn0s", "xlv1", "pi", "rslopermax", "rslopesmax", "rsloperbmax", "rslopesbmax",
               "rsloper2max", "rslopes2max", "rsloper3max", "rslopes3max"]
# what mp_driver.f90:105 / :554-585 pass (icar_constants.f90, wrf_constants.f90): den0, denr, dens, cliq, cpv
WSM3_INIT_ARGS = (1.28, 1000.0, 100.0, 4190.0, 4.0 * 461.6)


def wsm3_init(den0=WSM3_INIT_ARGS[0], denr=WSM3_INIT_ARGS[1], dens=WSM3_INIT_ARGS[2], cl=WSM3_INIT_ARGS[3], cpv=WSM3_INIT_ARGS[4]):
    out = np.zeros(len(WSM3_CONSTS), np.float32)
    lib().orc_wsm3_init(_f(den0), _f(denr), _f(dens), _f(cl), _f(np.float32(cpv)), _p(out))
    return dict(zip(WSM3_CONSTS, out))


def wsm3(th, q, qci, qrs, w, den, pii, p, delz, args18, rain, rainncv, snow, snowncv, sr, its, ite, jts, jte, kts, kte):
    ny, nz, nx = q.shape
    a = np.ascontiguousarray(args18, np.float32)
    fn = lib().orc_wsm3; fn.restype = ctypes.c_int
    return int(fn(_i(nx), _i(nz), _i(ny), _p(th), _p(q), _p(qci), _p(qrs), _p(w), _p(den), _p(pii), _p(p), _p(delz), _p(a),
                  _p(rain), _p(rainncv), _p(snow), _p(snowncv), _p(sr), *[_i(x) for x in (its, ite, jts, jte, kts, kte)]))


# ---- WSM6 (oracle/wsm6_oracle.c) -------------------------------------------------------------------------------------
WSM6_CONSTS = ("qc0 qck1 bvtr1 bvtr2 bvtr3 bvtr4 g1pbr g3pbr g4pbr g5pbro2 pvtr eacrr pacrr bvtr6 g6pbr precr1 precr2 roqimax bvts1 bvts2 "
               "bvts3 bvts4 g1pbs g3pbs g4pbs g5pbso2 pvts pacrs precs1 precs2 pidn0r pidn0s xlv1 pacrc pi bvtg1 bvtg2 bvtg3 bvtg4 g1pbg "
               "g3pbg g4pbg g5pbgo2 pvtg pacrg precg1 precg2 pidn0g rslopermax rslopesmax rslopegmax rsloperbmax rslopesbmax rslopegbmax "
               "rsloper2max rslopes2max rslopeg2max rsloper3max rslopes3max rslopeg3max").split()


def wsm6_init(den0=WSM3_INIT_ARGS[0], denr=WSM3_INIT_ARGS[1], dens=WSM3_INIT_ARGS[2], cl=WSM3_INIT_ARGS[3], cpv=WSM3_INIT_ARGS[4]):
    """wsm6init with the arguments of mp_driver.f90:100 (the same five as wsm3init's)"""
    out = np.zeros(len(WSM6_CONSTS), np.float32)
    lib().orc_wsm6_init(_f(den0), _f(denr), _f(dens), _f(cl), _f(np.float32(cpv)), _p(out))
    return dict(zip(WSM6_CONSTS, out))


def wsm6(th, q, qc, qr, qi, qs, qg, den, pii, p, delz, args18, rain, sr, snow, graupel, its, ite, jts, jte, kts, kte):
    ny, nz, nx = q.shape
    a = np.ascontiguousarray(args18, np.float32)
    fn = lib().orc_wsm6; fn.restype = ctypes.c_int
    return int(fn(_i(nx), _i(nz), _i(ny), _p(th), _p(q), _p(qc), _p(qr), _p(qi), _p(qs), _p(qg), _p(den), _p(pii), _p(p), _p(delz), _p(a),
                  _p(rain), _p(sr), _p(snow), _p(graupel), *[_i(x) for x in (its, ite, jts, jte, kts, kte)]))
