"""Builds tests/support/libicar_probe.so (th_probe.hip: the level code's device math functions evaluated on arrays; mpdata_probe.hip:
zeroes the antidiffusive coefficients of a context so that the fused MPDATA kernel returns its donor-cell pass) with the
product's own compile flags.  TEST INFRASTRUCTURE: called by __graft_entry__.build() (so that the file travels to the GPU box with
the other built libraries) and by the tests' `probe` fixture.  Cross-compiles without a GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libicar_probe.so")


def build(force=False):
    sys.path.insert(0, ROOT)
    from icar_amd import build as B
    srcs = [os.path.join(HERE, "th_probe.hip"), os.path.join(HERE, "mpdata_probe.hip")]
    deps = srcs + [os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith((".h", ".inc"))]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call([B.HIPCC] + B.FLAGS + ["-I" + B.CSRC, "-I" + os.path.join(ROOT, "include"), "-shared"] + srcs + ["-o", LIB])
    return LIB


def lib():
    L = ctypes.CDLL(build())
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.icar_probe_math.argtypes = [ci, ci, vp, vp, vp]
    L.icar_probe_dec_index.argtypes = [vp, vp, ci, ci, ci, vp]
    L.icar_probe_mpdata_zero_antidiffusion.argtypes = [vp]
    return L


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
