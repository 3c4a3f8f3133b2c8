"""options_t mirror: the members of src/objects/options_h.f90:12-53 / opt_types.f90 that the hot
path reads, with the reference's defaults (src/objects/options_obj.f90)."""
from dataclasses import dataclass, field
import numpy as np
from .constants import ADVECTION_ORDER, kADV_MPDATA, kMP_THOMPSON, kMP_SB04


@dataclass
class physics_type:                      # opt_types.f90:15-24
    microphysics: int = 0
    advection: int = kADV_MPDATA
    windtype: int = 0


@dataclass
class adv_options_type:                  # opt_types.f90:101-105, defaults options_obj.f90:1579-1581
    mpdata_order: int = 2
    flux_corrected_transport: bool = True
    boundary_buffer: bool = False


@dataclass
class mp_options_type:                   # opt_types.f90:30-46, defaults options_obj.f90:1259-1284
    Nt_c: float = 100.e6; TNO: float = 5.0; am_s: float = 0.069; rho_g: float = 500.0
    av_s: float = 40.0; bv_s: float = 0.55; fv_s: float = 100.0; av_g: float = 442.0; bv_g: float = 0.89
    av_i: float = 1847.5; Ef_si: float = 0.05; Ef_rs: float = 0.95; Ef_rg: float = 0.75; Ef_ri: float = 0.95
    C_cubes: float = 0.5; C_sqrd: float = 0.3; mu_r: float = 0.0; t_adjust: float = 0.0
    Ef_rw_l: bool = False; Ef_sw_l: bool = False
    update_interval: int = 0; top_mp_level: int = 0; local_precip_fraction: float = 1.0

    def as_arrays(self):
        p = np.array([self.Nt_c, self.TNO, self.am_s, self.rho_g, self.av_s, self.bv_s, self.fv_s, self.av_g,
                      self.bv_g, self.av_i, self.Ef_si, self.Ef_rs, self.Ef_rg, self.Ef_ri, self.C_cubes,
                      self.C_sqrd, self.mu_r, self.t_adjust], np.float32)
        return p, np.array([int(self.Ef_rw_l), int(self.Ef_sw_l)], np.int32)


@dataclass
class lt_options_type:                   # opt_types.f90 lt_options_type, defaults options_obj.f90:1447-1482
    buffer: int = 50
    stability_window_size: int = 10
    vert_smooth: int = 10
    variable_N: bool = True
    smooth_nsq: bool = True
    max_stability: float = 6e-4
    min_stability: float = 1e-7
    N_squared: float = 3e-5
    linear_contribution: float = 1.0
    linear_update_fraction: float = 0.2
    spatial_linear_fields: bool = True
    dirmax: float = float(np.float32(2) * np.float32(3.1415927))
    dirmin: float = 0.0
    spdmax: float = 30.0
    spdmin: float = 0.0
    nsqmax: float = None                 # log(max_stability) unless given (options_obj.f90:1474)
    nsqmin: float = None                 # log(min_stability)
    n_dir_values: int = 24
    n_nsq_values: int = 5
    n_spd_values: int = 6
    minimum_layer_size: float = 100.0
    read_LUT: bool = False
    write_LUT: bool = True

    def resolved(self):
        """(nsqmin, nsqmax) in REAL(4): log() of the REAL(4) stability limits like the namelist defaults."""
        import math
        hi = np.float32(self.nsqmax) if self.nsqmax is not None else np.float32(math.log(float(np.float32(self.max_stability))))
        lo = np.float32(self.nsqmin) if self.nsqmin is not None else np.float32(math.log(float(np.float32(self.min_stability))))
        return float(lo), float(hi)


@dataclass
class parameter_options_type:            # opt_types.f90:188-326 (subset on the path)
    dx: float = 1000.0
    dz_levels: np.ndarray = None
    advect_density: bool = False         # options_obj.f90:1023
    fixed_dz_advection: bool = True
    cfl_reduction_factor: float = 0.9    # options_obj.f90:1050
    cfl_strictness: int = 3              # options_obj.f90:1051
    debug: bool = False
    ideal: bool = False
    space_varying_dz: bool = False       # options_obj.f90:1936
    wind_iterations: int = 100           # options_obj.f90:1029
    restart_file: str = ""               # options_obj.f90 restart_info namelist
    restart_step_in_file: int = 1


@dataclass
class options_t:
    physics: physics_type = field(default_factory=physics_type)
    adv_options: adv_options_type = field(default_factory=adv_options_type)
    mp_options: mp_options_type = field(default_factory=mp_options_type)
    lt_options: lt_options_type = field(default_factory=lt_options_type)
    parameters: parameter_options_type = field(default_factory=parameter_options_type)
    vars_to_advect: dict = field(default_factory=dict)
    vars_to_allocate: dict = field(default_factory=dict)
    vars_for_restart: dict = field(default_factory=dict)

    # options_h.f90: alloc_vars / advect_vars / restart_vars OR requests into the masks
    def alloc_vars(self, names):
        for n in names: self.vars_to_allocate[n] = self.vars_to_allocate.get(n, 0) + 1

    def advect_vars(self, names):
        for n in names:
            if n not in ADVECTION_ORDER:
                raise ValueError(f"{n} is not an advectable kVARS entry")
            self.vars_to_advect[n] = self.vars_to_advect.get(n, 0) + 1

    def restart_vars(self, names):
        for n in names: self.vars_for_restart[n] = self.vars_for_restart.get(n, 0) + 1
