"""Shared synthetic inputs for the linear-wind tests (rows W2/W3)."""
import numpy as np


def terrain(nxg, nyg, seed=0, height=1200.0):
    """[nyg, nxg] float32 (== Fortran (nx,ny)): two Gaussian ridges + small roughness."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(nyg), np.arange(nxg), indexing="ij")
    t = height * np.exp(-(((x - 0.4 * nxg) / (0.12 * nxg)) ** 2 + ((y - 0.55 * nyg) / (0.2 * nyg)) ** 2))
    t += 0.5 * height * np.exp(-(((x - 0.75 * nxg) / (0.08 * nxg)) ** 2 + ((y - 0.3 * nyg) / (0.1 * nyg)) ** 2))
    t += 20.0 * rng.random((nyg, nxg))
    return t.astype(np.float32)


def lut_options(lt):
    lo, hi = lt.resolved()
    return dict(dirmin=lt.dirmin, dirmax=lt.dirmax, spdmin=lt.spdmin, spdmax=lt.spdmax, nsqmin=lo, nsqmax=hi,
                n_dir_values=lt.n_dir_values, n_spd_values=lt.n_spd_values, n_nsq_values=lt.n_nsq_values,
                minimum_layer_size=lt.minimum_layer_size)


def atmosphere(nx, ny, nz, seed=1, moist=True):
    """Fields spatial_winds reads, C-order (ny,nz,nx)."""
    rng = np.random.default_rng(seed)
    dz = np.linspace(60.0, 400.0, nz).astype(np.float32)
    zc = (np.cumsum(dz) - dz / 2).astype(np.float32)
    z = (zc[None, :, None] + 30.0 * rng.random((ny, 1, nx))).astype(np.float32)
    th = (285.0 + 0.004 * z + 0.3 * rng.standard_normal((ny, nz, nx))).astype(np.float32)
    p = (1e5 * np.exp(-z / 8000.0)).astype(np.float32)
    exner = ((p / np.float32(1e5)) ** np.float32(0.2856)).astype(np.float32)
    qv = (0.008 * np.exp(-z / 2500.0) * (1 + 0.1 * rng.standard_normal((ny, nz, nx)))).astype(np.float32)
    f = dict(z=z, potential_temperature=th, exner=exner, water_vapor=np.abs(qv), dz=dz)
    if moist:
        cloud = (rng.random((ny, nz, nx)) < 0.3)
        f["cloud_water_mass"] = (cloud * 4e-4 * rng.random((ny, nz, nx))).astype(np.float32)
        f["cloud_ice_mass"] = (cloud * 1e-8 * rng.random((ny, nz, nx))).astype(np.float32)
        f["rain_mass"] = ((rng.random((ny, nz, nx)) < 0.1) * 2e-4 * rng.random((ny, nz, nx))).astype(np.float32)
        f["snow_mass"] = np.zeros((ny, nz, nx), np.float32)
    f["u"] = (8.0 + 6.0 * rng.standard_normal((ny, nz, nx + 1))).astype(np.float32)
    f["v"] = (-3.0 + 6.0 * rng.standard_normal((ny + 1, nz, nx))).astype(np.float32)
    return f
