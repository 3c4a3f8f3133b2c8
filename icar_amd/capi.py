"""ctypes binding of libicar_hip.so (include/icar_hip.h).  No fallback: if the library is not
built, or no gfx950 device is visible, every compute entry point raises IcarHipError."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICAR_HIP_LIB") or os.path.join(_HERE, "lib", "libicar_hip.so")     # override: A/B builds of profiles/micro
_lib = None


class IcarHipError(RuntimeError):
    pass


# every symbol include/icar_hip.h declares (tests/test_abi.py cross-checks this list with the header)
SYMBOLS = [
    "icar_hip_ctx_create", "icar_hip_ctx_destroy", "icar_hip_set_stream", "icar_hip_synchronize",
    "icar_hip_aux_fork", "icar_hip_aux_begin", "icar_hip_aux_end", "icar_hip_aux_join", "icar_hip_max_courant_device",
    "icar_hip_field_upload", "icar_hip_field_download", "icar_hip_field_fill", "icar_hip_field_device_ptr",
    "icar_hip_field_count", "icar_hip_field_elem_size", "icar_hip_setup_winds", "icar_hip_advect",
    "icar_hip_mpdata_exact", "icar_hip_mp_simple", "icar_hip_mp_simple_tiles", "icar_hip_thompson_init", "icar_hip_thompson", "icar_hip_thompson_tiles", "icar_hip_thompson_table", "icar_hip_mp_tiles",
    "icar_hip_wsm3_init", "icar_hip_wsm3", "icar_hip_wsm6_init", "icar_hip_wsm6", "icar_hip_wsm6_tiles", "icar_hip_winds_valid", "icar_hip_max_courant", "icar_hip_max_courant_prefetch", "icar_hip_max_abs_winds", "icar_hip_balance_uvw", "icar_hip_balance_uvw_update", "icar_hip_make_winds_grid_relative", "icar_hip_mass_conservative_acceleration", "icar_hip_update_winds", "icar_hip_exchange_uv", "icar_hip_iterative_winds_correct_w", "icar_hip_iterative_winds_sweep", "icar_hip_box_pack", "icar_hip_box_unpack", "icar_hip_dqdt_download", "icar_hip_diagnostic_update", "icar_hip_diagnostic_update_parts", "icar_hip_dqdt_upload",
    "icar_hip_apply_forcing", "icar_hip_enforce_limits", "icar_hip_halo_count", "icar_hip_halo_pack",
    "icar_hip_halo_unpack", "icar_hip_halo_pack_dirs", "icar_hip_halo_unpack_dirs", "icar_hip_timing_enable", "icar_hip_timing_groups", "icar_hip_timing_read", "icar_hip_timing_reset",
    "icar_hip_last_error", "icar_hip_version",
    "icar_hip_comm_unique_id", "icar_hip_comm_init", "icar_hip_comm_init_host", "icar_hip_comm_destroy", "icar_hip_comm_kind", "icar_hip_comm_ranks", "icar_hip_comm_timeout", "icar_hip_halo_selfcheck",
    "icar_hip_halo_send", "icar_hip_halo_retrieve", "icar_hip_co_min", "icar_hip_co_max",
    "icar_hip_step_configure", "icar_hip_model_time_set", "icar_hip_model_time", "icar_hip_mp_reset", "icar_hip_compute_dt",
    "icar_hip_update_dt", "icar_hip_mp", "icar_hip_advect_step", "icar_hip_substep", "icar_hip_step", "icar_hip_step_n",
    "icar_hip_linwinds_setup", "icar_hip_linwinds_terrain_frequency", "icar_hip_linear_perturbation",
    "icar_hip_linwinds_build_lut", "icar_hip_linwinds_build_lut_varying", "icar_hip_linwinds_lut_download", "icar_hip_linwinds_lut_upload", "icar_hip_linwinds_lut_entry",
    "icar_hip_linwinds_perturbation_download", "icar_hip_linwinds_perturbation_upload", "icar_hip_spatial_winds",
]


class lt_options_c(ctypes.Structure):          # struct icar_hip_lt_options (include/icar_hip.h)
    _fields_ = [("buffer", ctypes.c_int), ("stability_window_size", ctypes.c_int), ("vert_smooth", ctypes.c_int),
                ("variable_N", ctypes.c_int), ("smooth_nsq", ctypes.c_int),
                ("max_stability", ctypes.c_float), ("min_stability", ctypes.c_float), ("N_squared", ctypes.c_float),
                ("linear_contribution", ctypes.c_float), ("linear_update_fraction", ctypes.c_float),
                ("dirmax", ctypes.c_float), ("dirmin", ctypes.c_float), ("spdmax", ctypes.c_float), ("spdmin", ctypes.c_float),
                ("nsqmax", ctypes.c_float), ("nsqmin", ctypes.c_float),
                ("n_dir_values", ctypes.c_int), ("n_nsq_values", ctypes.c_int), ("n_spd_values", ctypes.c_int),
                ("minimum_layer_size", ctypes.c_float)]


N_ADVECTABLE = 11
NEIGHBOR_NONE, NEIGHBOR_SELF = -1, -2
COMM_NONE, COMM_LOCAL, COMM_RCCL, COMM_HOST = 0, 1, 2, 3


class step_config_c(ctypes.Structure):         # struct icar_hip_step_config (include/icar_hip.h)
    _fields_ = [("advection", ctypes.c_int), ("microphysics", ctypes.c_int), ("mpdata_order", ctypes.c_int),
                ("flux_corrected_transport", ctypes.c_int), ("advect_density", ctypes.c_int), ("cfl_strictness", ctypes.c_int),
                ("cfl_reduction_factor", ctypes.c_float), ("dx", ctypes.c_float), ("mp_update_interval", ctypes.c_float),
                ("top_mp_level", ctypes.c_int), ("halo_size", ctypes.c_int),
                ("its", ctypes.c_int), ("ite", ctypes.c_int), ("jts", ctypes.c_int), ("jte", ctypes.c_int),
                ("kts", ctypes.c_int), ("kte", ctypes.c_int), ("ids", ctypes.c_int), ("ide", ctypes.c_int),
                ("jds", ctypes.c_int), ("jde", ctypes.c_int), ("kds", ctypes.c_int), ("kde", ctypes.c_int),
                ("west_boundary", ctypes.c_int), ("east_boundary", ctypes.c_int), ("south_boundary", ctypes.c_int),
                ("north_boundary", ctypes.c_int), ("diagnostics", ctypes.c_int), ("prefetch_dt", ctypes.c_int),
                ("n_advect", ctypes.c_int), ("advect_fields", ctypes.c_int * N_ADVECTABLE),
                ("n_exchange", ctypes.c_int), ("exchange_fields", ctypes.c_int * N_ADVECTABLE),
                ("n_forced", ctypes.c_int), ("forced_fields", ctypes.c_int * 16), ("force_boundaries", ctypes.c_int * 16)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IcarHipError(f"{LIB_PATH} is missing: run `python -m icar_amd.build` "
                               "(there is no CPU fallback for the hot path)")
        L = ctypes.CDLL(LIB_PATH)
        L.icar_hip_last_error.restype = ctypes.c_char_p
        L.icar_hip_version.restype = ctypes.c_char_p
        L.icar_hip_field_count.restype = ctypes.c_size_t
        L.icar_hip_field_elem_size.restype = ctypes.c_size_t
        L.icar_hip_halo_count.restype = ctypes.c_size_t
        L.icar_hip_field_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.icar_hip_halo_count.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.icar_hip_model_time.restype = cd
        L.icar_hip_model_time.argtypes = [vp]
        L.icar_hip_model_time_set.argtypes = [vp, cd]
        L.icar_hip_mp.argtypes = [vp, cd, ci, ci]
        L.icar_hip_advect_step.argtypes = [vp, cd]
        L.icar_hip_substep.argtypes = [vp, cd, ci]
        L.icar_hip_step.argtypes = [vp, cd, ctypes.POINTER(ci)]
        L.icar_hip_step_n.argtypes = [vp, ci, ctypes.POINTER(cd)]
        L.icar_hip_update_dt.argtypes = [vp, ctypes.POINTER(cd)]
        L.icar_hip_compute_dt.argtypes = [vp, ctypes.POINTER(cd)]
        L.icar_hip_co_min.argtypes = [vp, ctypes.POINTER(cd)]
        L.icar_hip_co_max.argtypes = [vp, ctypes.POINTER(cd)]
        L.icar_hip_step_configure.argtypes = [vp, ctypes.POINTER(step_config_c), vp]
        L.icar_hip_comm_init.argtypes = [vp, ci, ci, ctypes.c_char_p, ctypes.POINTER(ci)]
        L.icar_hip_comm_init_host.argtypes = [vp, ci, ci, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ci)]
        L.icar_hip_comm_unique_id.argtypes = [ctypes.c_char_p]
        L.icar_hip_halo_send.argtypes = [vp, ci, ctypes.POINTER(ci), ci]
        L.icar_hip_halo_retrieve.argtypes = [vp, ci, ctypes.POINTER(ci), ci]
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise IcarHipError(f"{what}: {lib().icar_hip_last_error().decode()}")
