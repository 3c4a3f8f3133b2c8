import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


_LIBM = {}


def host_libm_matches_device_math():
    """The device's REAL(4) expf / logf / log10f / powf / atanf (icar_amd/csrc/glibc_flt32.h) restate the FMA builds of glibc 2.35;
    the microphysics parity tests compare the device with the CPU oracle in "mode 0" = whatever libm.so.6 THIS host has.  On a
    host with another glibc, or without AVX2 + FMA (glibc then selects other builds), those tests would go red for a reason that
    has nothing to do with the kernels.  Returns (ok, reason): the restated functions compiled for the CPU against the host's libm
    on a sample (every 4096th REAL(4) bit pattern, 2e5 powf pairs; tests/glibc_flt32_check.cpp -- the full sweep is
    tests/test_glibc_flt32_host.py)."""
    if _LIBM:
        return _LIBM["ok"], _LIBM["why"]
    import subprocess, tempfile
    ok, why = True, ""
    try:
        flags = open("/proc/cpuinfo").read()
        if " fma" not in flags or " avx2" not in flags:
            ok, why = False, "no AVX2 + FMA on this host: its glibc selects the non-FMA builds of expf / logf / powf"
        else:
            exe = os.path.join(tempfile.mkdtemp(prefix="icar_libm_"), "check")
            subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "glibc_flt32_check.cpp"), "-o", exe])
            out = subprocess.check_output([exe, "4096", "200000"], text=True, timeout=300)
            bad = {l.split()[0]: int(l.split()[2]) for l in out.splitlines() if len(l.split()) >= 3}
            if any(bad.values()) or len(bad) < 5:
                ok, why = False, f"the host libm differs from the glibc 2.35 FMA builds icar_amd/csrc/glibc_flt32.h restates: {bad}"
            else:       # ... and the DOUBLE PRECISION exp / log / pow of the Thompson level code (glibc_dbl64.h), 2e5 arguments per class
                exe2 = exe + "_d"
                subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "glibc_dbl64_check.cpp"), "-o", exe2])
                out = subprocess.check_output([exe2, "200000"], text=True, timeout=300)
                bad = {l.split()[0]: int(l.split()[2]) for l in out.splitlines() if len(l.split()) >= 3 and not l.startswith(" ")}
                if any(bad.values()) or len(bad) < 12:
                    ok, why = False, f"the host libm differs from the glibc 2.35 FMA builds icar_amd/csrc/glibc_dbl64.h restates: {bad}"
    except Exception as e:  # no compiler, no libm ...: cannot tell -> do not hide the tests
        ok, why = True, f"(libm probe unavailable: {e})"
    _LIBM.update(ok=ok, why=why)
    return ok, why


def pytest_collection_modifyitems(config, items):
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU in this container")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)
        return
    ok, why = host_libm_matches_device_math()
    if ok:
        return
    # LOUD: every device-vs-oracle test (they take the `oracle` / `th_oracle` fixture) is skipped with the reason, and said once more at the end
    skip = pytest.mark.skip(reason="HOST LIBM MISMATCH, device-vs-oracle parity not checkable here: " + why)
    for item in items:
        if "gpu" in item.keywords and ({"oracle", "th_oracle"} & set(getattr(item, "fixturenames", ()))):
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    if _LIBM and not _LIBM["ok"]:
        terminalreporter.section("HOST LIBM MISMATCH")
        terminalreporter.write_line("device-vs-oracle GPU parity tests were SKIPPED on this host: " + _LIBM["why"])


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def th_oracle(oracle):
    """the oracle with the Thompson tables of the default mp_options built (thompson_init)"""
    from icar_amd.options import options_t
    p, f = options_t().mp_options.as_arrays()
    oracle.thompson_init(p, f)
    return oracle
