"""icar_amd/csrc/glibc_dbl64.h (the device's DOUBLE PRECISION exp / log / pow = glibc 2.35's FMA builds restated) compiled for the
CPU and compared with the host C library value by value: tests/glibc_dbl64_check.cpp.  In the suite: 2e7 arguments per class (13
classes: exp over its range / near 0 / all bit patterns, log over all binades / near 1 / all bit patterns, pow as the microphysics
uses it, quarter-integer exponents, all bit patterns, positive bases, results near over- and underflow) and a grid of special
values; `./check 1000000000` runs 1e9 per class (recorded in profiles/r04_parity.json: 0 mismatches)."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_double_functions_equal_libm(tmp_path):
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "glibc_dbl64_check.cpp"), "-o", exe])
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags or " avx2" not in flags:
        pytest.skip("this host's glibc selects the non-FMA builds of exp / log / pow")
    out = subprocess.check_output([exe, "20000000"], text=True, timeout=900)
    seen = {}
    for line in out.splitlines():
        if line.startswith(" "):
            continue
        name, n, bad = line.split()[:3]
        seen[name] = (int(n), int(bad))
        assert int(bad) == 0, out
    assert len(seen) == 14 and "libm_pow_one_is_x" in seen and "pow_physics" in seen and "special" in seen, out
