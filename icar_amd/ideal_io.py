"""Ideal-case input files: read the two NetCDF files an ICAR ideal run starts from into the members
of domain_t, and write them (the committed fixture generator uses the writer).

The reference's helpers write them with xarray as NetCDF-4 (`tests/gen_ideal_test.py:21` -> helpers/genNetCDF/Topography.py:76-83,
Forcing.py:59-81); the image has no HDF5 library, so both sides here are NetCDF CLASSIC through scipy.io.netcdf_file with the
SAME variable names, dimension names and orders -- the format contract the reference's own readers key on
(io_routines.f90 reads by variable name; options `hgt_hi`, `lat_hi`, `lon_hi`; forcing `uvar`, `vvar`, `pvar`, `tvar`, `qvvar`,
`zvar`, `latvar`, `lonvar`):
    init.nc     lat_hi(lat, lon), lon_hi(lat, lon), hgt_hi(lat, lon)                              [Topography.py:62-69]
    forcing.nc  u, v, theta, qv, z, pressure, temperature (time, level, lat, lon); height(lat, lon);
                lat_m(lat), lon_m(lon), x_m(x_m), time(time)                                       [Forcing.py:62-79]
What the reference does between these files and domain_t (geographic look-up tables, vertical interpolation, the SLEVE
coordinate: domain_obj.f90 ~2000 lines, boundary_obj.f90) is init-time code outside the hot path; the reader
here implements its ideal-case subset: nearest forcing column by lat / lon (the ideal forcing is horizontally uniform apart from
the Schaer blob), linear interpolation in height, the terrain-following coordinate of icar_amd.ideal.  Parity unpinned."""
import numpy as np
from scipy.io import netcdf_file
from . import ideal
from ._netcdf import open_classic, FORMAT_NOTE

G, RD, CP = 9.81, 287.058, 1003.5                       # Forcing.py:352-356


def calc_wk_theta(z):
    """Weisman-Klemp potential temperature profile (Forcing.py:330-346)."""
    z = np.asarray(z, np.float64)
    lo = 300.0 + (343.0 - 300.0) * (np.minimum(z, 12000.0) / 12000.0) ** 1.25
    hi = 343.0 * np.exp((G / (1000.0 * 213.0)) * (z - 12000.0))
    return np.where(z <= 12000.0, lo, hi)


def calc_pressure_from_sea(p0, z):
    """Forcing.py:364-365"""
    return p0 * (1 - 2.25577e-5 * np.asarray(z, np.float64)) ** 5.25588


def lat_lon(nx, ny, dx, dy, lat0, lon0):
    """Topography.py:48-57 / Forcing.py:29-36: 111,111 m per degree, centred on (lat0, lon0)."""
    dlon = dx / 111111 / np.cos(np.radians(lat0)); dlat = dy / 111111
    lon = np.arange(lon0 - nx / 2 * dlon, lon0 + nx / 2 * dlon, dlon)[:nx]
    lat = np.arange(lat0 - ny / 2 * dlat, lat0 + ny / 2 * dlat, dlat)[:ny]
    return lat, lon


def hills(nx, ny, hill_height, n_hills):
    """Topography.py:139-165: one cosine hill (n_hills == 1) or the cos^2 exp mountain range (n_hills > 1); 0 -> flat."""
    i = (np.arange(nx) - nx / 2) / nx * np.pi * 2
    j = (np.arange(ny) - ny / 2) / ny * np.pi * 2
    ig, jg = np.meshgrid(i, j)
    if n_hills == 0:
        return np.zeros((ny, nx))
    if n_hills == 1:
        return ((np.cos(ig) + 1) * (np.cos(jg) + 1)) / 4 * hill_height
    c, sigma = 0.15, n_hills ** 2
    return (np.cos(ig / c) ** 2 * np.exp(-(ig / c) ** 2 / sigma) * np.cos(jg / c) ** 2 * np.exp(-(jg / c) ** 2 / sigma)) * hill_height


def write_init(path, nx, ny, dx=1000.0, dy=1000.0, hill_height=1000.0, n_hills=1, lat0=39.5, lon0=-105.0):
    lat, lon = lat_lon(nx, ny, dx, dy, lat0, lon0)
    lon2, lat2 = np.meshgrid(lon, lat)
    with netcdf_file(path, "w", version=2) as f:
        f.format_note = FORMAT_NOTE
        f.TITLE = "OUTPUT FROM CONTINUOUS INTEGRATION TEST"; f.GRIDTYPE = "C"; f.DX = float(dx); f.DY = float(dy)   # Topography.py:213-222
        f.createDimension("lat", ny); f.createDimension("lon", nx)
        for name, arr, units, desc in (("lat_hi", lat2, "degrees latitude", "Latitude on mass grid"),
                                       ("lon_hi", lon2, "degrees longitude", "Longitude on mass grid"),
                                       ("hgt_hi", hills(nx, ny, hill_height, n_hills), "meters MSL", "topography height")):
            v = f.createVariable(name, "d", ("lat", "lon")); v[:] = arr; v.units = units; v.description = desc


def write_forcing(path, nt, nz, nx, ny, dz_value=500.0, dx=1000.0, dy=1000.0, u_val=10.0, v_val=0.0, qv_val=0.001,
                  sealevel_pressure=100000.0, lat0=39.5, lon0=-105.0):
    """Forcing.py with weather_model 'WeismanKlemp' and pressure_func 'calc_pressure_from_sea' (gen_ideal_test.py:69-74)."""
    lat, lon = lat_lon(nx, ny, dx, dy, lat0, lon0)
    z1 = np.cumsum(np.concatenate([[0.0], np.full(nz - 1, dz_value)]))                 # Forcing.py:176-189
    shape = (nt, nz, ny, nx)
    col = lambda prof: np.broadcast_to(np.asarray(prof, np.float64)[None, :, None, None], shape)
    u_prof = np.full(nz, u_val) if np.isscalar(u_val) else np.asarray(u_val, np.float64)[:nz]
    theta = calc_wk_theta(z1); p = calc_pressure_from_sea(sealevel_pressure, z1)
    temp = theta * (p / 100000.0) ** (RD / CP)                                          # Forcing.py:382-383
    with netcdf_file(path, "w", version=2) as f:
        f.format_note = FORMAT_NOTE
        f.createDimension("time", None); f.createDimension("level", nz); f.createDimension("lat", ny); f.createDimension("lon", nx)
        f.createDimension("x_m", nx)
        d4 = ("time", "level", "lat", "lon")
        for name, arr, long_name, units in (("u", col(u_prof), "U (E/W) wind speed", "m s**-1"), ("v", np.full(shape, v_val), "V (N/S) wind speed", "m s**-1"),
                                            ("theta", col(theta), "Potential Temperature", "K"), ("qv", np.full(shape, qv_val), "Relative Humidity", "kg kg**-1"),
                                            ("z", col(z1), "Atmospheric Elevation", "m"), ("pressure", col(p), "Pressure", "Pa"),
                                            ("temperature", col(temp), "Temperature", "K")):
            v = f.createVariable(name, "d", d4); v[:] = arr; v.long_name = long_name; v.units = units
        v = f.createVariable("height", "d", ("lat", "lon")); v[:] = 0.0; v.long_name = "Topographic Height"; v.units = "m"
        v = f.createVariable("lat_m", "d", ("lat",)); v[:] = lat; v.long_name = "latitude"; v.units = "degree_north"
        v = f.createVariable("lon_m", "d", ("lon",)); v[:] = lon; v.long_name = "longitude"; v.units = "degree_east"
        v = f.createVariable("x_m", "d", ("x_m",)); v[:] = np.arange(-nx * dx / 2, nx * dx / 2, dx)[:nx]; v.units = "meters"
        v = f.createVariable("time", "i", ("time",)); v[:] = np.arange(nt); v.units = "hours since 2020-12-01 00:00:00"   # Forcing.py:52-54, :81


def read_ideal_case(init_file, forcing_file, dz_levels, dx, time_index=0, n_hydro=0):
    """-> a case dict with the keys of icar_amd.ideal.make_case (domain_t.load_case takes it): terrain from hgt_hi; the forcing
    column nearest in lat / lon to every cell, interpolated linearly in height to the mass levels (clamped beyond the forcing's
    range); exner / density from the interpolated pressure and potential temperature; u, v staggered from the mass-point winds;
    w from balance_uvw.  dz_levels = options%parameters%dz_levels, dx = options%parameters%dx."""
    f32 = np.float32
    with open_classic(init_file) as f:
        terrain = np.array(f.variables["hgt_hi"][:], f32)
        lat_hi = np.array(f.variables["lat_hi"][:]); lon_hi = np.array(f.variables["lon_hi"][:])
    ny, nx = terrain.shape
    nz = len(dz_levels)
    dzl32 = np.asarray(dz_levels, f32)
    uniform = float(dzl32[0]) if np.all(dzl32 == dzl32[0]) else None
    if uniform is None and not np.array_equal(dzl32, ideal.dz_levels(nz)):
        raise ValueError("read_ideal_case: dz_levels must be uniform or the default level table (icar_amd.ideal.dz_levels)")
    case = ideal.make_case(nx, ny, nz, dx=dx, terrain=terrain, n_hydro=n_hydro, uniform_dz=uniform)     # the geometry: jacobians, dz
    with open_classic(forcing_file) as f:
        lat_m = np.array(f.variables["lat_m"][:]); lon_m = np.array(f.variables["lon_m"][:])
        fz = np.array(f.variables["z"][time_index]); fields = {k: np.array(f.variables[k][time_index]) for k in ("u", "v", "theta", "qv", "pressure")}
    # geographic look-up: nearest forcing column (geo.f90's bilinear weights reduce to it where the two grids coincide)
    jj = np.abs(lat_hi[:, :, None] - lat_m[None, None, :]).argmin(axis=2)
    ii = np.abs(lon_hi[:, :, None] - lon_m[None, None, :]).argmin(axis=2)
    dzl = case["dz_levels"].astype(np.float64)
    Hs = dzl.sum(); jac2d = (Hs - terrain.astype(np.float64)) / Hs
    z_if = np.concatenate([[0.0], np.cumsum(dzl)]); zc = 0.5 * (z_if[1:] + z_if[:-1])
    z = terrain[:, None, :].astype(np.float64) + zc[None, :, None] * jac2d[:, None, :]            # (ny, nz, nx) mass-level height

    def vinterp(name, log=False):
        src = fields[name][:, jj, ii]                         # (nz_f, ny, nx): the column of every cell
        zs = fz[:, jj, ii]
        out = np.empty((ny, nz, nx))
        for j in range(ny):
            for i in range(nx):
                col = np.log(src[:, j, i]) if log else src[:, j, i]
                o = np.interp(z[j, :, i], zs[:, j, i], col)
                out[j, :, i] = np.exp(o) if log else o
        return out
    theta = vinterp("theta"); p = vinterp("pressure", log=True); qv = vinterp("qv"); um = vinterp("u"); vm = vinterp("v")
    exner = (p / 1.0e5) ** (float(ideal.RD) / float(ideal.CP))
    case.update(potential_temperature=theta.astype(f32), pressure=p.astype(f32), water_vapor=qv.astype(f32), exner=exner.astype(f32),
                density=(p / (float(ideal.RD) * theta * exner)).astype(f32), z=z.astype(f32))
    u = np.empty((ny, nz, nx + 1), f32); v = np.empty((ny + 1, nz, nx), f32)
    u[:, :, 1:nx] = 0.5 * (um[:, :, 1:] + um[:, :, :-1]); u[:, :, 0] = um[:, :, 0]; u[:, :, nx] = um[:, :, nx - 1]
    v[1:ny] = 0.5 * (vm[1:] + vm[:-1]); v[0] = vm[0]; v[ny] = vm[ny - 1]
    case["u"], case["v"] = u, v
    case["w"] = ideal.balance_uvw(u, v, case["jacobian_u"], case["jacobian_v"], case["jacobian_w"], case["advection_dz"], float(dx))
    return case
