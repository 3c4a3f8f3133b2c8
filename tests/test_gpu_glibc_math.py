"""The device's REAL(4) exp / log / log10 / x**y / atan (icar_amd/csrc/glibc_flt32.h: the C library's expf / logf / log10f /
powf / atanf restated) against the HOST's libm, bit for bit, on millions of arguments per function -- including the shared-base
form of powf the Thompson level code uses and its 10.**x.  The same header is checked on the CPU against every one of the 2^32
REAL(4) arguments in tests/test_glibc_flt32_host.py; this is the run on the device (v_fma_f64, v_cvt, denormal handling)."""
import ctypes
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.capi import lib, check
from util import single_image_domain, parity_record

pytestmark = pytest.mark.gpu


def test_device_float_transcendentals_equal_the_hosts_libm(oracle, probe):
    rng = np.random.default_rng(2024)

    def run(op, x, y=None):
        x = np.ascontiguousarray(x, np.float64); out = np.zeros(x.size, np.float64)
        yp = None if y is None else np.ascontiguousarray(y, np.float64).ctypes.data_as(ctypes.c_void_p)
        assert probe.icar_probe_math(op, x.size, x.ctypes.data_as(ctypes.c_void_p), yp, out.ctypes.data_as(ctypes.c_void_p)) == 0, "math_probe"
        return out.astype(np.float32)

    def same(a, b):
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))

    n = 3_000_000
    anybits = lambda m: rng.integers(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-40, 1.1754944e-38, 3.4028235e38, -3.4028235e38,
                        88.0, 88.7, 88.73, -87.3, -87.4, -103.0, -103.5, -104.0, -110.0, 0.4375, 0.6875, 1.1875, 2.4375, 3e7, 4e7], np.float32)
    stats = {}
    if True:
        one_arg = {4: ("expf", np.concatenate([rng.uniform(-104.0, 89.0, n).astype(np.float32), anybits(n // 3), special])),
                   5: ("logf", np.concatenate([np.abs(anybits(n)), (1.0 + rng.uniform(-1e-3, 1e-3, n // 3)).astype(np.float32), anybits(n // 10), special])),
                   6: ("log10f", np.concatenate([np.abs(anybits(n)), (10.0 ** rng.uniform(-8, 8, n // 3)).astype(np.float32), special])),
                   7: ("atanf", np.concatenate([anybits(n), rng.uniform(-4.0, 4.0, n).astype(np.float32), special])),
                   9: ("10**x", np.concatenate([rng.uniform(-46.0, 39.0, n).astype(np.float32), special]))}
        for op, (name, x) in one_arg.items():
            got, want = run(op, x), oracle.libm_f(op, x)
            bad = ~same(got, want)
            stats[name] = {"n": int(x.size), "differ": int(bad.sum())}
            assert not bad.any(), f"{name}: {bad.sum()} of {x.size} differ from libm, first x = {x[bad][0]!r}: {got[bad][0]!r} vs {want[bad][0]!r}"
        # powf: the scheme's use (positive base from any binade, moderate exponent), arbitrary bit patterns, special pairs
        xb = np.concatenate([np.abs(anybits(n)), (10.0 ** rng.uniform(-12.0, 12.0, n)).astype(np.float32), anybits(n // 2)])
        yb = np.concatenate([rng.uniform(-12.0, 12.0, n).astype(np.float32), rng.uniform(-6.0, 6.0, n).astype(np.float32), anybits(n // 2)])
        sx, sy = np.meshgrid(special, special)
        xb = np.concatenate([xb, sx.ravel()]); yb = np.concatenate([yb, sy.ravel()])
        want = oracle.libm_f(3, xb, yb)
        for op, name in ((3, "powf"), (8, "powf_shared_base")):
            got = run(op, xb, yb)
            bad = ~same(got, want)
            stats[name] = {"n": int(xb.size), "differ": int(bad.sum())}
            assert not bad.any(), f"{name}: {bad.sum()} of {xb.size} differ from libm, first ({xb[bad][0]!r}, {yb[bad][0]!r}): {got[bad][0]!r} vs {want[bad][0]!r}"
        parity_record("glibc_math", "device REAL(4) transcendentals vs the host libm (bit patterns)", stats)


EXHAUSTIVE = bool(__import__("os").environ.get("ICAR_EXHAUSTIVE"))


def test_device_one_argument_functions_on_every_real4(oracle, probe):
    """expf, logf, log10f, atanf on the device against the host's libm over the WHOLE REAL(4) line: by default a stratified sample of
    2^24 bit patterns (every 256th, with a per-function offset: every exponent, both signs, denormals, infinities and NaNs are in
    it; seconds), with ICAR_EXHAUSTIVE=1 all 2^32 (minutes; run once per round, result in profiles/r0N_parity.json)."""
    stride = 1 if EXHAUSTIVE else 256
    chunk = 1 << 25 if EXHAUSTIVE else 1 << 24
    stats = {}
    out = np.zeros(chunk, np.float64)
    for op, name in ((4, "expf"), (5, "logf"), (6, "log10f"), (7, "atanf")):
        differ = 0; first = None; n = 0
        off = 0 if EXHAUSTIVE else (37 * op) % stride
        for lo in range(0, 1 << 32, chunk * stride):
            x = (np.arange(chunk, dtype=np.uint64) * stride + (lo + off)).astype(np.uint32).view(np.float32)
            xd = x.astype(np.float64)
            assert probe.icar_probe_math(op, chunk, xd.ctypes.data_as(ctypes.c_void_p), None, out.ctypes.data_as(ctypes.c_void_p)) == 0
            got, want = out.astype(np.float32), oracle.libm_f(op, x)
            bad = ~((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))
            nb = int(bad.sum())
            if nb and first is None:
                first = (hex(int(x[bad][0].view(np.uint32))), float(got[bad][0]), float(want[bad][0]))
            differ += nb; n += chunk
        stats[name] = {"n": n, "differ": differ, "first": first}
        assert differ == 0, f"{name}: {differ} of {n} differ from libm, first {first}"
    parity_record("glibc_math_exhaustive" if EXHAUSTIVE else "glibc_math_sweep",
                  "device expf / logf / log10f / atanf vs the host libm on " + ("all 2^32" if EXHAUSTIVE else "a stratified 2^24 sample of the") + " REAL(4) arguments", stats)


def test_device_powf_on_many_pairs(oracle, probe):
    """powf (and the level code's shared-base form) on the device against the host's libm: half of the pairs arbitrary bit patterns,
    half positive bases of every binade with exponents in +-16 (the scheme's use).  2^24 pairs by default, 2^30 with ICAR_EXHAUSTIVE=1."""
    chunk = 1 << 22 if not EXHAUSTIVE else 1 << 24
    nchunks = 4 if not EXHAUSTIVE else 64
    rng = np.random.default_rng(77)
    stats = {"powf": {"n": 0, "differ": 0}, "powf_shared_base": {"n": 0, "differ": 0}}
    first = None
    out = np.zeros(chunk, np.float64)
    for it in range(nchunks):
        if it % 2 == 0:
            x = rng.integers(0, 2 ** 32, chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
            y = rng.integers(0, 2 ** 32, chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
        else:
            x = np.abs(rng.integers(0, 2 ** 32, chunk, dtype=np.uint64).astype(np.uint32).view(np.float32))
            y = rng.uniform(-16.0, 16.0, chunk).astype(np.float32)
        want = oracle.libm_f(3, x, y)
        xd, yd = x.astype(np.float64), y.astype(np.float64)
        for op, name in ((3, "powf"), (8, "powf_shared_base")):
            assert probe.icar_probe_math(op, chunk, xd.ctypes.data_as(ctypes.c_void_p), yd.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
            got = out.astype(np.float32)
            bad = ~((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))
            nb = int(bad.sum())
            if nb and first is None:
                first = (name, float(x[bad][0]), float(y[bad][0]), float(got[bad][0]), float(want[bad][0]))
            stats[name]["n"] += chunk; stats[name]["differ"] += nb
    assert stats["powf"]["differ"] == 0 and stats["powf_shared_base"]["differ"] == 0, (stats, first)
    parity_record("glibc_math_exhaustive" if EXHAUSTIVE else "glibc_math_sweep", f"device powf vs the host libm on {stats['powf']['n']} pairs", stats)
