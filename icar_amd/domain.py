"""domain_t mirror: device-resident fields of one tile behind the reference's member names
(src/objects/domain_h.f90:18-363), plus halo_send / halo_retrieve / halo_exchange
(src/objects/domain_obj.f90:109-143 over src/objects/exchangeable_obj.f90:138-356).

Device memory lives in libicar_hip's context; numpy arrays only cross at upload/download
(forcing / output boundaries), like the reference's NetCDF I/O points."""
import ctypes
import numpy as np
from . import _fields as F
from .capi import lib, check, IcarHipError
from .constants import ADVECTION_ORDER, KVARS
from .grid import grid_t

DIR_NORTH, DIR_SOUTH, DIR_EAST, DIR_WEST = 0, 1, 2, 3


class domain_t:
    def __init__(self, grid: grid_t, device=0, dx=1000.0, image=1, comm=None):
        self.grid = grid
        self.dx = float(dx)
        self.image = image
        self.comm = comm                 # halo transport (icar_amd.halo); None for a single image
        for n in ("ims", "ime", "jms", "jme", "kms", "kme", "its", "ite", "jts", "jte", "kts", "kte",
                  "ids", "ide", "jds", "jde", "kds", "kde"):
            setattr(self, n, getattr(grid, n))
        self._ctx = ctypes.c_void_p()
        check(lib().icar_hip_ctx_create(ctypes.byref(self._ctx), int(device), grid.ims, grid.ime, grid.kms,
                                        grid.kme, grid.jms, grid.jme), "icar_hip_ctx_create")
        self.nx, self.nz, self.ny = grid.ime - grid.ims + 1, grid.kme - grid.kms + 1, grid.jme - grid.jms + 1
        self.device = int(device)
        self.model_time_seconds = 0.0    # domain%model_time%seconds()
        self.mp_state = dict(last_model_time=-999.0)   # mp_driver.f90's SAVE variables last_model_time / update_interval
        self.exchange_vars = []          # kVARS names with an associated exchangeable (halo_send order)

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def ctx(self):
        if not self._ctx:
            raise IcarHipError("domain context destroyed")
        return self._ctx

    def close(self):
        if self._ctx:
            lib().icar_hip_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shape(self, fid):
        if fid in (F.U, F.JACOBIAN_U, F.DZDX, F.ZR_U): return (self.ny, self.nz, self.nx + 1)
        if fid in (F.V, F.JACOBIAN_V, F.DZDY, F.ZR_V): return (self.ny + 1, self.nz, self.nx)
        if fid in F.IS_2DD or fid in (F.SURFACE_PRESSURE, F.IVT, F.IWV, F.IWL, F.IWI): return (self.ny, self.nx)
        return (self.ny, self.nz, self.nx)

    @staticmethod
    def fid(name):
        if isinstance(name, int): return name
        if name in F.NAMES: return F.NAMES[name]
        if name in KVARS: return KVARS[name][0]
        raise KeyError(name)

    def set(self, name, array):
        """Upload a host array (C-order (ny,nz,nx) == Fortran (i,k,j)) into the named member."""
        fid = self.fid(name)
        dt = np.float64 if fid in F.IS_2DD else np.float32
        a = np.ascontiguousarray(array, dtype=dt)
        if a.shape != self.shape(fid):
            raise ValueError(f"{name}: shape {a.shape} != {self.shape(fid)}")
        check(lib().icar_hip_field_upload(self.ctx, fid, a.ctypes.data_as(ctypes.c_void_p)), f"upload {name}")
        if fid in (F.SINTHETA, F.COSTHETA):
            self._has_theta = True

    def get(self, name):
        fid = self.fid(name)
        a = np.empty(self.shape(fid), np.float64 if fid in F.IS_2DD else np.float32)
        check(lib().icar_hip_field_download(self.ctx, fid, a.ctypes.data_as(ctypes.c_void_p)), f"download {name}")
        return a

    def set_dqdt(self, name, array):
        """variable_t%dqdt_3d of a forced variable (same shape as the variable)."""
        fid = self.fid(name)
        a = np.ascontiguousarray(array, dtype=np.float32)
        if a.shape != self.shape(fid):
            raise ValueError(f"dqdt {name}: shape {a.shape} != {self.shape(fid)}")
        check(lib().icar_hip_dqdt_upload(self.ctx, fid, a.ctypes.data_as(ctypes.c_void_p)), f"dqdt_upload {name}")

    def get_dqdt(self, name):
        fid = self.fid(name)
        a = np.empty(self.shape(fid), np.float32)
        check(lib().icar_hip_dqdt_download(self.ctx, fid, a.ctypes.data_as(ctypes.c_void_p)), f"dqdt_download {name}")
        return a

    def apply_forcing(self, dt_seconds, forced):
        """domain%apply_forcing(dt) (domain_obj.f90:2383-2448). forced = [(name, force_boundaries), ...];
        include ("w", False) like the reference's separate w update."""
        ids = (ctypes.c_int * len(forced))(*[self.fid(n) for n, _ in forced])
        fb = (ctypes.c_int * len(forced))(*[int(bool(b)) for _, b in forced])
        g = self.grid
        check(lib().icar_hip_apply_forcing(self.ctx, ctypes.c_double(dt_seconds), ids, fb, len(forced), int(g.west_boundary),
                                           int(g.east_boundary), int(g.south_boundary), int(g.north_boundary)), "apply_forcing")

    def enforce_limits(self, names):
        """domain%enforce_limits (domain_obj.f90:2228-2243): clamp negatives to zero."""
        ids = (ctypes.c_int * len(names))(*[self.fid(n) for n in names])
        check(lib().icar_hip_enforce_limits(self.ctx, ids, len(names)), "enforce_limits")

    def prefetch_courant(self, options):
        """The strictness-3 / 4 CFL reduction of the NEXT update_dt, launched now on the current stream (time_step.substep puts it
        on the second stream beside the advection); the library hands it out only if nothing writes u, v, w in between."""
        dzl = np.ascontiguousarray(options.parameters.dz_levels, np.float32)
        check(lib().icar_hip_max_courant_prefetch(self.ctx, ctypes.c_float(self.dx), dzl.ctypes.data_as(ctypes.c_void_p)),
              "icar_hip_max_courant_prefetch")

    def diagnostic_update(self, parts=3):
        """diagnostic_update(domain, options) (time_step.f90:49-198).  parts: 1 = all but w_real, 2 = w_real only, 3 = both."""
        check(lib().icar_hip_diagnostic_update_parts(self.ctx, int(parts)), "diagnostic_update")

    def fill(self, name, value):
        check(lib().icar_hip_field_fill(self.ctx, self.fid(name), ctypes.c_double(value)), f"fill {name}")

    def device_ptr(self, name):
        p = ctypes.c_void_p()
        check(lib().icar_hip_field_device_ptr(self.ctx, self.fid(name), ctypes.byref(p)), f"device_ptr {name}")
        return p.value

    def synchronize(self):
        check(lib().icar_hip_synchronize(self.ctx), "synchronize")

    # second HIP stream (include/icar_hip.h: aux_fork / aux_begin / aux_end / aux_join)
    def aux_fork(self):
        check(lib().icar_hip_aux_fork(self.ctx), "aux_fork")

    def aux_begin(self):
        check(lib().icar_hip_aux_begin(self.ctx), "aux_begin")

    def aux_end(self):
        check(lib().icar_hip_aux_end(self.ctx), "aux_end")

    def aux_join(self):
        check(lib().icar_hip_aux_join(self.ctx), "aux_join")

    def set_stream(self, stream_ptr):
        """Run the context's kernels on the caller's HIP stream (0 / None = the context's own non-blocking stream)."""
        check(lib().icar_hip_set_stream(self.ctx, ctypes.c_void_p(stream_ptr)), "set_stream")
        self._stream_ptr = int(stream_ptr) if stream_ptr else 0

    def bind_torch_stream(self, stream=None):
        """Put the context on a (non-default) torch stream and make it torch's current stream, so that RCCL's
        send/recv -- which order themselves against torch's current stream -- are ordered with the pack / unpack
        kernels without host synchronisation."""
        import torch
        if stream is None:
            stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        self.set_stream(stream.cuda_stream)
        self._torch_stream = stream
        return stream

    def needs_host_sync(self):
        """True when the context's stream is not torch's current stream (then HaloComm synchronises on the host)."""
        import torch
        cur = torch.cuda.current_stream().cuda_stream
        return not (getattr(self, "_stream_ptr", 0) and self._stream_ptr == cur)

    def load_case(self, case):
        """Upload every member present in an icar_amd.ideal case dict."""
        alias = {"cloud_water": "cloud_water_mass", "rain": "rain_mass", "snow": "snow_mass",
                 "cloud_ice": "cloud_ice_mass", "graupel": "graupel_mass", "ice_number": "cloud_ice_number"}
        for k, v in case.items():
            name = alias.get(k, k)
            if isinstance(v, np.ndarray) and name in F.NAMES and v.ndim >= 2:
                self.set(name, v)

    # ---- halo exchange (H1) ----------------------------------------------------------------
    def _exchange_fields(self):
        order = {n: i for i, n in enumerate(ADVECTION_ORDER)}
        return [KVARS[n][0] for n in sorted(self.exchange_vars, key=lambda n: order[n])]

    def halo_send(self):
        """domain_obj.f90:109-128: put my edge planes of every exchangeable to the 4 neighbours."""
        if self.comm is not None:
            self.comm.send(self, self._exchange_fields())

    def halo_retrieve(self):
        """domain_obj.f90:130-143: sync with neighbours, then copy the inboxes into my halo planes."""
        if self.comm is not None:
            self.comm.retrieve(self, self._exchange_fields())

    def halo_exchange(self):
        self.halo_send()
        self.halo_retrieve()


# ---- tile interface used by icar_amd.halo.HaloComm (device buffers are torch tensors) ------------
def _halo_count(self, direction, halo):
    return int(lib().icar_hip_halo_count(self.ctx, int(direction), int(halo)))


def _new_buffer(self, n):
    import torch
    return torch.empty(int(n), dtype=torch.float32, device=f"cuda:{self.device}")     # the context's device, not torch's current one


def _halo_pack(self, direction, halo, field_ids, buf):
    arr = (ctypes.c_int * len(field_ids))(*field_ids)
    check(lib().icar_hip_halo_pack(self.ctx, int(direction), int(halo), arr, len(field_ids),
                                   ctypes.c_void_p(buf.data_ptr())), "icar_hip_halo_pack")


def _halo_unpack(self, direction, halo, field_ids, buf):
    arr = (ctypes.c_int * len(field_ids))(*field_ids)
    check(lib().icar_hip_halo_unpack(self.ctx, int(direction), int(halo), arr, len(field_ids),
                                     ctypes.c_void_p(buf.data_ptr())), "icar_hip_halo_unpack")


def _halo_many(self, fn, what, directions, halo, field_ids, bufs):
    arr = (ctypes.c_int * len(field_ids))(*field_ids)
    dirs = (ctypes.c_int * len(directions))(*[int(x) for x in directions])
    ptrs = (ctypes.c_void_p * len(bufs))(*[ctypes.c_void_p(b.data_ptr()) for b in bufs])
    check(fn(self.ctx, len(directions), dirs, int(halo), arr, len(field_ids), ptrs), what)


def _halo_pack_many(self, directions, halo, field_ids, bufs):
    """all directions of a halo_send in one launch (icar_hip_halo_pack_dirs)"""
    _halo_many(self, lib().icar_hip_halo_pack_dirs, "icar_hip_halo_pack_dirs", directions, halo, field_ids, bufs)


def _halo_unpack_many(self, directions, halo, field_ids, bufs):
    _halo_many(self, lib().icar_hip_halo_unpack_dirs, "icar_hip_halo_unpack_dirs", directions, halo, field_ids, bufs)


def _box_pack(self, field, which, i0, ni, j0, nj, buf):
    check(lib().icar_hip_box_pack(self.ctx, int(field), int(which), int(i0), int(ni), int(j0), int(nj),
                                  ctypes.c_void_p(buf.data_ptr())), "icar_hip_box_pack")


def _box_unpack(self, field, which, i0, ni, j0, nj, buf):
    check(lib().icar_hip_box_unpack(self.ctx, int(field), int(which), int(i0), int(ni), int(j0), int(nj),
                                    ctypes.c_void_p(buf.data_ptr())), "icar_hip_box_unpack")


domain_t.box_pack = _box_pack
domain_t.box_unpack = _box_unpack
domain_t.halo_count = _halo_count
domain_t.new_buffer = _new_buffer
domain_t.halo_pack = _halo_pack
domain_t.halo_unpack = _halo_unpack
domain_t.halo_pack_many = _halo_pack_many
domain_t.halo_unpack_many = _halo_unpack_many
