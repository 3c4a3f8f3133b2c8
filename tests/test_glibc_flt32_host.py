"""icar_amd/csrc/glibc_flt32.h (the device's expf / logf / log10f / powf / atanf) compiled for the CPU and compared with the
host C library value by value: tests/glibc_flt32_check.cpp.  In the suite: every 16th REAL(4) bit pattern of the one-argument
functions (268 M arguments each) and 4e7 argument pairs of powf; `./check 1 1000000000` runs all 2^32 + 1e9 (about a minute on
8 cores; 0 mismatches recorded in profiles/r03_parity.json)."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_float_functions_equal_libm(tmp_path):
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "glibc_flt32_check.cpp"), "-o", exe])
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags or " avx2" not in flags:
        pytest.skip("this host's glibc selects the non-FMA builds of expf / logf / powf")
    out = subprocess.check_output([exe, "16", "40000000"], text=True, timeout=900)
    seen = {}
    for line in out.splitlines():
        name, n, bad = line.split()[:3]
        seen[name] = (int(n), int(bad))
        assert int(bad) == 0, line
    assert set(seen) == {"expf", "logf", "log10f", "atanf", "powf"} and all(n > 1e7 for n, _ in seen.values()), out
