"""The C-ABI shared library loads without a GPU and exports every symbol include/icar_hip.h
declares; context creation fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import pytest
from icar_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "icar_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(icar_hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        from icar_amd import build
        build.build()
    L = capi.lib()
    for s in header_symbols():
        assert hasattr(L, s), f"{s} not exported by libicar_hip.so"
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert set(header_symbols()) <= exported


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = ctypes.c_void_p()
    rc = capi.lib().icar_hip_ctx_create(ctypes.byref(c), 0, 1, 16, 1, 8, 1, 16)
    assert rc != 0 and not c.value
    assert b"no HIP device" in capi.lib().icar_hip_last_error()
    with pytest.raises(capi.IcarHipError):
        from icar_amd.domain import domain_t
        from icar_amd.grid import grid_t
        domain_t(grid_t().set_grid_dimensions(16, 16, 8, 1, 1))


def test_mp_tiles_partition_the_tile():
    """M0 (mp_driver.f90:609-658): halo ring U interior(subset) == tile, disjoint; integer-exact."""
    from icar_amd.microphysics import mp_tiles
    import numpy as np
    for (its, ite, jts, jte, h) in [(2, 99, 2, 99, 1), (257, 511, 2, 256, 1), (5, 40, 7, 19, 2), (1, 6, 1, 5, 1)]:
        cover = np.zeros((jte + 2, ite + 2), int)
        for (a, b, c_, d) in mp_tiles(its, ite, jts, jte, halo=h) + mp_tiles(its, ite, jts, jte, subset=h):
            cover[c_:d + 1, a:b + 1] += 1
        assert (cover[jts:jte + 1, its:ite + 1] == 1).all() and cover.sum() == (ite - its + 1) * (jte - jts + 1)
    assert mp_tiles(2, 99, 2, 99, halo=1) == [(2, 2, 2, 99), (99, 99, 2, 99), (3, 98, 2, 2), (3, 98, 99, 99)]
