"""icar_amd: MI355X (gfx950) implementation of ICAR's per-timestep 3-D grid update behind the
reference's domain_t / options_t operator interface.  See DESIGN.md / INTEGRATION.md."""
from .capi import IcarHipError, lib, LIB_PATH          # noqa: F401
from .grid import grid_t                               # noqa: F401
from .options import options_t                         # noqa: F401
from .constants import kADV_UPWIND, kADV_MPDATA, kMP_THOMPSON, kMP_SB04   # noqa: F401


def __getattr__(name):
    if name == "domain_t":
        from .domain import domain_t
        return domain_t
    raise AttributeError(name)
