"""domain_t mirror: device-resident fields of one tile behind the reference's member names
(src/objects/domain_h.f90:18-363), plus halo_send / halo_retrieve / halo_exchange
(src/objects/domain_obj.f90:109-143 over src/objects/exchangeable_obj.f90:138-356).

Device memory lives in libicar_hip's context; numpy arrays only cross at upload/download
(forcing / output boundaries), like the reference's NetCDF I/O points."""
import ctypes
import numpy as np
from . import _fields as F
from .capi import lib, check, IcarHipError, step_config_c, NEIGHBOR_NONE, NEIGHBOR_SELF
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
        self.exchange_vars = []          # kVARS names with an associated exchangeable (halo_send order)
        self._step_key = None            # what the library's step driver was last configured with (configure())
        self._forced, self._diagnostics, self._prefetch_dt = (), True, True
        if comm is not None:
            comm.attach(self)            # icar_hip_comm_init[_host]: collective over the images of the communicator

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def ctx(self):
        if not self._ctx:
            raise IcarHipError("domain context destroyed")
        return self._ctx

    # domain%model_time%seconds(): the clock lives in the library (the step driver and mp()'s update_interval gating read it)
    @property
    def model_time_seconds(self):
        return float(lib().icar_hip_model_time(self.ctx))

    @model_time_seconds.setter
    def model_time_seconds(self, seconds):
        check(lib().icar_hip_model_time_set(self.ctx, float(seconds)), "icar_hip_model_time_set")

    def configure(self, options, forced=None, diagnostics=None, prefetch_dt=None, advection=None):
        """icar_hip_step_configure: hand the library the members of options_t / grid_t that step(), update_dt(), mp() and
        advect() read (time_step.f90:440-551).  Cheap when nothing changed.  forced = [(member, force_boundaries), ...] (the
        variables apply_forcing updates), diagnostics (diagnostic_update at the top of a sub-step) and prefetch_dt stay as last
        given when omitted.  advection=0 overrides options%physics%advection (time_step.mp_and_halo: the microphysics + halo
        block alone)."""
        p, g = options.parameters, self.grid
        adv = options.physics.advection if advection is None else advection
        adv_ids = tuple(KVARS[n][0] for n in ADVECTION_ORDER if options.vars_to_advect.get(n, 0) > 0)
        if forced is not None: self._forced = tuple((self.fid(n), int(bool(b))) for n, b in forced)
        if diagnostics is not None: self._diagnostics = bool(diagnostics)
        if prefetch_dt is not None: self._prefetch_dt = bool(prefetch_dt)
        forced, diagnostics, prefetch_dt = self._forced, self._diagnostics, self._prefetch_dt
        dzl = p.dz_levels
        key = (adv, options.physics.microphysics, int(options.adv_options.mpdata_order), bool(options.adv_options.flux_corrected_transport),
               bool(p.advect_density), int(p.cfl_strictness), float(p.cfl_reduction_factor), float(options.mp_options.update_interval),
               int(options.mp_options.top_mp_level), adv_ids, tuple(self._exchange_fields()), forced, bool(diagnostics), bool(prefetch_dt),
               id(dzl), None if dzl is None else float(np.asarray(dzl, np.float32).sum()))
        if key == self._step_key:
            return
        c = step_config_c()
        c.advection, c.microphysics = int(adv), int(options.physics.microphysics)
        c.mpdata_order, c.flux_corrected_transport = int(options.adv_options.mpdata_order), int(bool(options.adv_options.flux_corrected_transport))
        c.advect_density, c.cfl_strictness = int(bool(p.advect_density)), int(p.cfl_strictness)
        c.cfl_reduction_factor, c.dx = float(p.cfl_reduction_factor), self.dx
        c.mp_update_interval, c.top_mp_level = float(options.mp_options.update_interval), int(options.mp_options.top_mp_level)
        c.halo_size = int(getattr(self.comm, "halo", None) or g.halo_size)
        for n in ("its", "ite", "jts", "jte", "kts", "kte", "ids", "ide", "jds", "jde", "kds", "kde"):
            setattr(c, n, int(getattr(g, n)))
        for n in ("west_boundary", "east_boundary", "south_boundary", "north_boundary"):
            setattr(c, n, int(bool(getattr(g, n))))
        c.diagnostics, c.prefetch_dt = int(bool(diagnostics)), int(bool(prefetch_dt))
        c.n_advect = len(adv_ids)
        for m, f in enumerate(adv_ids): c.advect_fields[m] = f
        ex = self._exchange_fields()
        c.n_exchange = len(ex)
        for m, f in enumerate(ex): c.exchange_fields[m] = f
        c.n_forced = len(forced)
        for m, (f, b) in enumerate(forced): c.forced_fields[m] = f; c.force_boundaries[m] = b
        dz = np.ascontiguousarray(dzl if dzl is not None else np.ones(self.nz), np.float32)
        if dz.shape != (self.nz,):
            raise ValueError(f"options%parameters%dz_levels has {dz.shape[0]} levels, the tile {self.nz}")
        check(lib().icar_hip_step_configure(self.ctx, ctypes.byref(c), dz.ctypes.data_as(ctypes.c_void_p)), "icar_hip_step_configure")
        self._step_key = key

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

    def _halo_args(self):
        ids = self._exchange_fields()
        return int(self.comm.halo), (ctypes.c_int * len(ids))(*ids), len(ids)

    def halo_send(self):
        """domain_obj.f90:109-128: put my edge planes of every exchangeable to the 4 neighbours (icar_hip_halo_send: one pack
        launch + one RCCL send/recv group on the context's stream)."""
        if self.comm is not None:
            check(lib().icar_hip_halo_send(self.ctx, *self._halo_args()), "icar_hip_halo_send")

    def halo_retrieve(self):
        """domain_obj.f90:130-143: sync with neighbours, then copy the inboxes into my halo planes (icar_hip_halo_retrieve)."""
        if self.comm is not None:
            check(lib().icar_hip_halo_retrieve(self.ctx, *self._halo_args()), "icar_hip_halo_retrieve")

    def co_min(self, value):
        """`call co_min(seconds)` (time_step.f90:413) over the images of this domain's communicator."""
        v = ctypes.c_double(float(value))
        check(lib().icar_hip_co_min(self.ctx, ctypes.byref(v)), "icar_hip_co_min")
        return v.value

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
