"""The Fortran 2008 host (icar_amd/fortran/icar_hip_mod.f90 + icar_hip_demo.f90, built by flang in
build()) drives the device hot path through the C ABI: mp_simple + MPDATA for 3 steps on an ideal
hill tile, compared bit-for-bit with the CPU oracle (device-math mode) on the same inputs."""
import os
import subprocess
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd import build as b

pytestmark = pytest.mark.gpu


def test_fortran_host_matches_oracle(oracle, tmp_path):
    demo = b.DEMO if os.path.exists(b.DEMO) else b.build_fortran_host()
    if not demo or not os.path.exists(demo):
        pytest.skip("flang not available to build the Fortran host")
    nx, ny, nz, nsteps = 48, 30, 16, 3
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.01)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.6)).astype(np.float32)
    dt = min(ideal.cfl_dt(c), 40.0)
    names = ["w", "pressure", "exner", "density", "dz_mass", "jacobian", "jacobian_w", "advection_dz", "water_vapor", "cloud_water",
             "rain", "snow", "potential_temperature", "u", "v", "jacobian_u", "jacobian_v"]
    for n in names:
        c[n].tofile(tmp_path / f"{n}.bin")
    (tmp_path / "meta.txt").write_text(f"{nx} {nz} {ny} {nsteps} {dt!r} {float(c['dx'])!r}\n")
    r = subprocess.run([demo, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "icar_hip_demo: ok" in r.stdout, r.stdout + r.stderr
    dt = float(np.float32(dt))
    # oracle: same operator sequence
    s = {k: c[k].copy() for k in ["pressure", "potential_temperature", "exner", "density", "water_vapor", "cloud_water", "rain", "snow", "dz_mass"]}
    acc = np.zeros((ny, nx), np.float64)
    order = ["water_vapor", "cloud_water", "rain", "snow", "potential_temperature"]
    oracle.set_math_mode(1)
    try:
        for _ in range(nsteps):
            rain = np.zeros((ny, nx), np.float32); snow = rain.copy()
            oracle.mp_simple(s["pressure"], s["potential_temperature"], s["exner"], s["density"], s["water_vapor"], s["cloud_water"],
                             s["rain"], s["snow"], rain, snow, dt, s["dz_mass"], 2, nx - 1, 2, ny - 1, 1, nz)
            acc += rain
            q = np.stack([s[n] for n in order]).copy()
            oracle.advect(2, q, c["u"], c["v"], c["w"], c["density"], c["jacobian"], c["jacobian_u"], c["jacobian_v"], c["jacobian_w"],
                          c["advection_dz"], c["dz_levels"], float(c["dx"]), dt)
            for m, n in enumerate(order): s[n] = q[m].copy()
    finally:
        oracle.set_math_mode(0)
    for n in order:
        got = np.fromfile(tmp_path / f"out_{n}.bin", np.float32).reshape(ny, nz, nx)
        assert np.array_equal(got, s[n]), f"{n}: {(got != s[n]).sum()} cells differ"
    got = np.fromfile(tmp_path / "out_precip.bin", np.float64).reshape(ny, nx)
    assert np.array_equal(got, acc) and acc.max() > 0
