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
class parameter_options_type:            # opt_types.f90:188-326 (subset on the path)
    dx: float = 1000.0
    dz_levels: np.ndarray = None
    advect_density: bool = False         # options_obj.f90:1023
    fixed_dz_advection: bool = True
    cfl_reduction_factor: float = 0.9    # options_obj.f90:1050
    cfl_strictness: int = 3              # options_obj.f90:1051
    debug: bool = False
    ideal: bool = False


@dataclass
class options_t:
    physics: physics_type = field(default_factory=physics_type)
    adv_options: adv_options_type = field(default_factory=adv_options_type)
    mp_options: mp_options_type = field(default_factory=mp_options_type)
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
