"""Synthetic ideal-case inputs for the hot path (host side, numpy, init-time only).

The reference's ideal cases are produced by Python helpers that cannot run here
(helpers/gen_init_ideal.py:38-58, tests/gen_ideal_test.py:21, helpers/genNetCDF/Topography.py
`genHill`, helpers/genNetCDF/Forcing.py:335-369); this module restates their analytic formulas
(SURVEY.md section 8(c),(d)):

* 2-D cosine hill   hgt = ((cos ig + 1)(cos jg + 1))/4 * H
* Weisman-Klemp     theta(z) = 300 + 43 (z/12000)^1.25 below 12 km
* barometric        p = 1e5 (1 - 2.25577e-5 z)^5.25588
* stretched levels  dz = [50,75,125,200,300,400] + [500]*...
* Gal-Chen jacobian (H_s - terrain)/H_s              (src/objects/domain_obj.f90:1249)

Arrays are float32, C-order (ny, nz, nx) == Fortran (i,k,j) with i fastest; u is (ny,nz,nx+1),
v is (ny+1,nz,nx).
"""
import numpy as np

RD = np.float32(287.058)      # src/constants/icar_constants.f90:391
CP = np.float32(1012.0)       # :393
GRAVITY = np.float32(9.81)


def dz_levels(nz, uniform=None):
    if uniform is not None:
        return np.full(nz, uniform, np.float32)
    base = [50.0, 75.0, 125.0, 200.0, 300.0, 400.0]
    lev = (base + [500.0] * max(0, nz - len(base)))[:nz]
    return np.asarray(lev, np.float32)


def cosine_hill(nx, ny, height):
    ig = (np.arange(nx, dtype=np.float64) / max(nx - 1, 1) * 2 - 1) * np.pi
    jg = (np.arange(ny, dtype=np.float64) / max(ny - 1, 1) * 2 - 1) * np.pi
    h = ((np.cos(ig)[None, :] + 1) * (np.cos(jg)[:, None] + 1)) / 4 * height
    return h.astype(np.float32)


def sat_mr(t, p):
    """Saturated mixing ratio, same Magnus form as src/physics/mp_simple.f90:146-182."""
    t = np.asarray(t, np.float64); p = np.asarray(p, np.float64)
    a = np.where(t < 273.15, 21.8745584, 17.2693882)
    b = np.where(t < 273.15, 7.66, 35.86)
    es = 610.78 * np.exp(a * (t - 273.16) / (t - b))
    es = np.where(p - es <= 0, p * 0.99999, es)
    return 0.6219907 * es / (p - es)


def balance_uvw(u, v, jaco_u, jaco_v, jaco_w, dz, dx):
    """w such that div(u,v,w)=0; numpy statement of src/physics/wind.f90:81-169 (float32)."""
    um = u * jaco_u
    vm = v * jaco_v
    div = ((um[:, :, 1:] - um[:, :, :-1]) + (vm[1:, :, :] - vm[:-1, :, :])) / np.float32(dx)
    w = np.zeros_like(div)
    nz = div.shape[1]
    for k in range(nz):
        if k == 0:
            w[:, k, :] = np.float32(0) - div[:, k, :] * dz[:, k, :] / jaco_w[:, k, :]
        else:
            w[:, k, :] = (w[:, k - 1, :] * jaco_w[:, k - 1, :] - div[:, k, :] * dz[:, k, :]) / jaco_w[:, k, :]
    return w.astype(np.float32)


def make_case(nx, ny, nz, dx=1000.0, hill_height=0.0, u0=10.0, v0=3.0, uniform_dz=None,
              blob_amp=0.004, noise=0.0, seed=1234, n_hydro=0, cool=0.0, exact=False, terrain=None):
    """Build the synthetic state described in SURVEY.md section 8(d).

    exact=True builds the same kind of state from IEEE-exact operations only (+,-,*,/,sqrt; rational
    bumps instead of cos/exp/pow) so that the inputs are bit-reproducible on any host CPU/libm --
    used by the large golden fixtures whose inputs are regenerated rather than stored.

    Returns a dict of float32 arrays: u,v,w,jacobian,jacobian_u,jacobian_v,jacobian_w,
    advection_dz,dz_levels,dz_mass,pressure,exner,density,potential_temperature,water_vapor,
    cloud_water,rain,snow,cloud_ice,graupel,ice_number,rain_number (+ scalars dx).
    """
    f32 = np.float32
    dzl = dz_levels(nz, uniform_dz)
    if terrain is not None:
        terrain = np.ascontiguousarray(terrain, f32)
        assert terrain.shape == (ny, nx)
    elif hill_height > 0 and exact:
        xg = (np.arange(nx, dtype=np.float64) / max(nx - 1, 1) * 2 - 1); yg = (np.arange(ny, dtype=np.float64) / max(ny - 1, 1) * 2 - 1)
        bx = (1 - xg * xg) * (1 - xg * xg); by = (1 - yg * yg) * (1 - yg * yg)
        terrain = (by[:, None] * bx[None, :] * hill_height).astype(f32)
    else:
        terrain = cosine_hill(nx, ny, hill_height) if hill_height > 0 else np.zeros((ny, nx), f32)
    Hs = f32(dzl.sum())
    jac2d = ((Hs - terrain) / Hs).astype(f32)
    jaco = np.ascontiguousarray(np.broadcast_to(jac2d[:, None, :], (ny, nz, nx))).astype(f32)
    # staggered jacobians (src/objects/domain_obj.f90:1365-1385)
    jaco_u = np.empty((ny, nz, nx + 1), f32)
    jaco_u[:, :, 0] = jaco[:, :, 0]; jaco_u[:, :, nx] = jaco[:, :, nx - 1]
    jaco_u[:, :, 1:nx] = (jaco[:, :, 1:] + jaco[:, :, :-1]) / f32(2)
    jaco_v = np.empty((ny + 1, nz, nx), f32)
    jaco_v[0] = jaco[0]; jaco_v[ny] = jaco[ny - 1]
    jaco_v[1:ny] = (jaco[1:] + jaco[:-1]) / f32(2)
    jaco_w = np.empty((ny, nz, nx), f32)
    jaco_w[:, :-1, :] = (jaco[:, :-1, :] + jaco[:, 1:, :]) / f32(2)
    jaco_w[:, -1, :] = jaco[:, -1, :]
    adv_dz = np.ascontiguousarray(np.broadcast_to(dzl[None, :, None], (ny, nz, nx))).astype(f32)
    dz_mass = (adv_dz * jaco).astype(f32)

    u = np.full((ny, nz, nx + 1), u0, f32)
    v = np.full((ny + 1, nz, nx), v0, f32)
    w = balance_uvw(u, v, jaco_u, jaco_v, jaco_w, adv_dz, dx)

    # mass-level heights above sea level
    z_if = np.concatenate([[0.0], np.cumsum(dzl.astype(np.float64))])
    zc = 0.5 * (z_if[1:] + z_if[:-1])
    z = terrain[:, None, :].astype(np.float64) + zc[None, :, None] * jac2d[:, None, :]
    ii = np.arange(nx, dtype=np.float64)[None, None, :]
    jj = np.arange(ny, dtype=np.float64)[:, None, None]
    sig = max(nx / 8.0, 1.5)
    r2 = ((ii - nx / 2.0) * (ii - nx / 2.0) + (jj - ny / 2.0) * (jj - ny / 2.0)) / (2 * sig * sig)
    if exact:
        xz = np.minimum(z / 12000.0, 1.0)
        theta = 300.0 + 43.0 * xz * np.sqrt(np.sqrt(xz)) - cool
        b = 1.0 - z / 44330.0
        p = 1.0e5 * b * b * b * b * b
        exner = np.sqrt(np.sqrt(p / 1.0e5))
        T = theta * exner
        rho = p / (float(RD) * T)
        zs = z / 2000.0
        qv = 0.8 * 0.008 / (1.0 + zs * zs)
        blob = blob_amp / ((1.0 + r2) * (1.0 + r2))
        qv = qv + blob / (1.0 + z / 2500.0)
    else:
        theta = 300.0 + 43.0 * np.minimum(z / 12000.0, 1.0) ** 1.25
        theta = theta - cool
        p = 1.0e5 * (1.0 - 2.25577e-5 * z) ** 5.25588
        exner = (p / 1.0e5) ** (float(RD) / float(CP))
        T = theta * exner
        rho = p / (float(RD) * T)
        qsat = sat_mr(T, p)
        qv = 0.8 * qsat * np.where(z < 3000.0, 1.0, np.exp(-(z - 3000.0) / 2500.0))
        blob = blob_amp * np.exp(-r2)
        qv = qv + blob * np.exp(-z / 2500.0)
    if noise > 0:
        rng = np.random.default_rng(seed)
        qv = qv * (1.0 + noise * rng.uniform(-1, 1, qv.shape))
    case = dict(
        nx=nx, ny=ny, nz=nz, dx=f32(dx), dz_levels=dzl, terrain=terrain,
        u=u, v=v, w=w, jacobian=jaco, jacobian_u=jaco_u, jacobian_v=jaco_v, jacobian_w=jaco_w,
        advection_dz=adv_dz, dz_mass=dz_mass,
        pressure=p.astype(f32), exner=exner.astype(f32), density=rho.astype(f32),
        potential_temperature=theta.astype(f32), water_vapor=qv.astype(f32),
    )
    for name in ("cloud_water", "rain", "snow", "cloud_ice", "graupel", "ice_number", "rain_number"):
        case[name] = np.zeros((ny, nz, nx), f32)
    if n_hydro:
        # small positive hydrometeor fields so that every advected scalar has structure
        rng = np.random.default_rng(seed + 1)
        for name in ("cloud_water", "rain", "snow", "cloud_ice", "graupel"):
            case[name] = (1e-4 * rng.uniform(0, 1, (ny, nz, nx)) * (blob / max(blob_amp, 1e-12))).astype(f32)
        case["ice_number"] = (1e4 * rng.uniform(0, 1, (ny, nz, nx))).astype(f32)
        case["rain_number"] = (1e3 * rng.uniform(0, 1, (ny, nz, nx))).astype(f32)
    return {k: (np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a) for k, a in case.items()}


def cfl_dt(case, cfl=0.9):
    """dt from the strictness-3 rule of src/main/time_step.f90:264-289 (numpy, init-time)."""
    u, v, w = case["u"], case["v"], case["w"]
    dx = float(case["dx"])
    dzl = case["dz_levels"].astype(np.float64)
    mu = np.maximum(np.abs(u[:, :, :-1]), np.abs(u[:, :, 1:])) / dx
    mv = np.maximum(np.abs(v[:-1]), np.abs(v[1:])) / dx
    wl = np.concatenate([w[:, :1, :], w[:, :-1, :]], axis=1)
    mw = np.maximum(np.abs(w), np.abs(wl)) / dzl[None, :, None]
    return float(cfl / (mu + mv + mw).max())


def cut_tile(case, grid):
    """The part of a whole-domain case a grid_t tile holds in memory (ims..ime, jms..jme, halos included): what every image
    reads from the same input file in the reference (domain_obj.f90 read_domain_shape / setup).  Staggered members keep their
    extra column / row; scalars and 1-D members pass through."""
    g = grid
    nxg, nyg = case["nx"], case["ny"]
    tile = {}
    for k, v in case.items():
        if not isinstance(v, np.ndarray) or v.ndim < 2:
            tile[k] = v
        elif v.ndim == 2:
            tile[k] = np.ascontiguousarray(v[g.jms - 1:g.jme, g.ims - 1:g.ime])
        elif v.shape[2] == nxg + 1:
            tile[k] = np.ascontiguousarray(v[g.jms - 1:g.jme, :, g.ims - 1:g.ime + 1])
        elif v.shape[0] == nyg + 1:
            tile[k] = np.ascontiguousarray(v[g.jms - 1:g.jme + 1, :, g.ims - 1:g.ime])
        else:
            tile[k] = np.ascontiguousarray(v[g.jms - 1:g.jme, :, g.ims - 1:g.ime])
    tile["nx"], tile["ny"] = g.ime - g.ims + 1, g.jme - g.jms + 1
    return tile
