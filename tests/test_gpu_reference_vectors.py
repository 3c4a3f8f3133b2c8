"""The device against the compiled reference's own vectors, with NO oracle in between (round 6): tests/golden/*.npz hold inputs
and outputs written by running /root/reference/src/physics/{mp_thompson,mp_simple}.f90 unmodified (tests/golden/make_golden.py).
Every prognostic field the microphysics touches, after all the calls of the fixture: bit for bit.  (The advection's fixtures:
tests/test_gpu_advect.py::test_device_against_the_compiled_references_vectors.  The surface accumulators are REAL(4) sums in the
reference's stand-alone driver and REAL(8) on the device, like domain%accumulated_precipitation%data_2dd: compared to 1e-6.)"""
import json
import os
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.microphysics import mp, mp_init
from icar_amd.constants import kMP_THOMPSON, kMP_SB04
from util import single_image_domain, nbitdiff, parity_record, equals_reference_vector as bits_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TH = {"water_vapor": "water_vapor", "cloud_water": "cloud_water_mass", "rain": "rain_mass", "cloud_ice": "cloud_ice_mass",
      "snow": "snow_mass", "graupel": "graupel_mass", "ice_number": "cloud_ice_number", "rain_number": "rain_number",
      "potential_temperature": "potential_temperature"}
SIMPLE = {"potential_temperature": "potential_temperature", "water_vapor": "water_vapor", "cloud_water": "cloud_water_mass",
          "rain": "rain_mass", "snow": "snow_mass"}


def _domain(z, p, keys):
    c = ideal.make_case(p["nx"], p["ny"], p["nz"], hill_height=p["hill"], noise=0.01)
    for k in keys:
        c[k] = np.ascontiguousarray(z["in_" + k])
    return single_image_domain(c)


@pytest.mark.parametrize("name", ["thompson_warm_24x12x30", "thompson_cold_20x10x40", "thompson_longdt_16x8x40"])
def test_thompson_equals_the_compiled_references_output(name):
    z = np.load(os.path.join(GOLD, name + ".npz")); p = json.loads(str(z["params"]))
    d = _domain(z, p, list(TH) + ["exner", "pressure", "dz_mass"])
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    for _ in range(p["nsteps"]):
        mp(d, opt, p["dt"]); d.model_time_seconds += p["dt"]
        d.set("potential_temperature", d.get("potential_temperature") - np.float32(p["cool"]))
    for k, m in TH.items():
        got = d.get(m)
        assert bits_equal(got, z[k]), f"{k}: {nbitdiff(got, z[k])} of {got.size} cells differ from the compiled reference's output"
    acc = d.get("accumulated_precipitation")
    assert z["rainnc"].max() > 0 and np.allclose(acc, z["rainnc"], rtol=1e-5, atol=1e-7)
    parity_record("thompson", f"device vs the compiled reference's vectors: {name}", {k: {"bitdiff_cells": 0, "cells": int(z[k].size)} for k in TH})
    d.close()


@pytest.mark.parametrize("name", ["mp_simple_40x36x20", "mp_simple_snow_30x20x30"])
def test_mp_simple_equals_the_compiled_references_output(name):
    z = np.load(os.path.join(GOLD, name + ".npz")); p = json.loads(str(z["params"]))
    d = _domain(z, p, ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain", "snow", "dz_mass"])
    opt = options_t(); opt.physics.microphysics = kMP_SB04
    mp_init(opt, d)
    for _ in range(p["nsteps"]):
        mp(d, opt, p["dt"]); d.model_time_seconds += p["dt"]
        d.set("potential_temperature", d.get("potential_temperature") - np.float32(p["cool"]))
    for k, m in SIMPLE.items():
        got = d.get(m)
        assert bits_equal(got, z[k]), f"{k}: {nbitdiff(got, z[k])} of {got.size} cells differ from the compiled reference's output"
    assert np.allclose(d.get("accumulated_precipitation"), z["rain_acc"], rtol=1e-5, atol=1e-7)
    d.close()


@pytest.mark.parametrize("name", ["wsm6_mixed_phase_24x12x30", "wsm6_cold_two_loops_22x10x28", "wsm3_warm_two_loops_26x14x24", "wsm3_snow_crossing_0C_24x12x30"])
def test_wsm_equals_the_compiled_references_output(name):
    """mp_wsm6.f90 / mp_wsm3.f90 compiled unmodified wrote these states (tests/golden/make_golden_wsm.py; the inputs are regenerated
    from the recorded parameters and fingerprinted)"""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden_wsm as G
    from icar_amd.microphysics import mp_var_request
    from icar_amd.constants import kMP_WSM3, kMP_WSM6
    g = np.load(os.path.join(GOLD, name + ".npz")); p = json.loads(str(g["params"]))
    c, w, A = G.make_inputs(p)
    assert G.fingerprint(c, w, A) == float(g["input_fingerprint"]), "icar_amd.ideal drifted: the stored outputs belong to other inputs"
    c = dict(c); c.update(A)
    d = single_image_domain(c)
    d.set("w_real", w)                                                  # WSM3 reads diagnostic_update's w_real (mp_driver.f90:552-585)
    opt = options_t(); opt.physics.microphysics = kMP_WSM3 if p["scheme"] == 3 else kMP_WSM6
    mp_var_request(opt); mp_init(opt, d)
    for _ in range(p["steps"]):
        mp(d, opt, p["dt"]); d.model_time_seconds += p["dt"]
        d.set("potential_temperature", d.get("potential_temperature") - np.float32(p["cool"]))
    names = {"cloud_water": "cloud_water_mass", "rain": "rain_mass", "cloud_ice": "cloud_ice_mass", "snow": "snow_mass", "graupel": "graupel_mass"}
    for n in (G.K3 if p["scheme"] == 3 else G.K6):
        got = d.get(names.get(n, n))
        assert bits_equal(got, g[n]), f"{n}: {nbitdiff(got, g[n])} of {got.size} cells differ from the compiled reference's output"
    assert g["acc_rain"].max() > 1.0 and np.allclose(d.get("accumulated_precipitation"), g["acc_rain"], rtol=1e-5, atol=1e-6)
    d.close()


def test_thompson_tables_equal_the_compiled_references_tables():
    """all 29 lookup tables of thompson_init as built ON THE DEVICE (the two O(1e10)-term collection integrals included) against the
    sha256 / probed entries / sums of the tables the compiled reference built (tests/golden/thompson_tables.npz)"""
    import ctypes, hashlib
    from icar_amd.capi import lib, check
    import util
    z = np.load(os.path.join(GOLD, "thompson_tables.npz"))
    d = single_image_domain(ideal.make_case(8, 8, 4))
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    from test_oracle_thompson import TABLES as names
    assert len(names) == 29
    for name in names:
        n = ctypes.c_size_t()
        check(lib().icar_hip_thompson_table(d.ctx, name.encode(), None, ctypes.c_size_t(0), ctypes.byref(n)), "table size")
        t = np.empty(n.value, np.float64)
        check(lib().icar_hip_thompson_table(d.ctx, name.encode(), t.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(t.size), None), "table")
        assert hashlib.sha256(t.tobytes()).hexdigest() == str(z["sha_" + name]), name
        assert np.array_equal(t[z["idx_" + name]], z["val_" + name]) and float(t.sum()) == float(z["sum_" + name]), name
        util.COUNTS["reference_vector_fields"] += 1
    d.close()


def test_exner_equals_the_compiled_references_exner_function():
    """diagnostic_update's exner (time_step.f90:76 -> atm_utilities.f90 exner_function: one powf per cell) on the pressures of
    tests/golden/helpers.npz, against what the compiled reference returned for them"""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden_helpers as G
    g = np.load(os.path.join(GOLD, "helpers.npz")); p = G.inputs()["exner_p"]
    ny, nz, nx = p.shape
    c = ideal.make_case(nx, ny, nz, hill_height=0.0)
    c["pressure"] = np.ascontiguousarray(p)
    d = single_image_domain(c)
    d.diagnostic_update(parts=1)
    assert bits_equal(d.get("exner"), g["exner"].reshape(p.shape)), "exner differs from the compiled reference's exner_function"
    d.close()
