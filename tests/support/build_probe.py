"""Builds tests/support/libicar_probe.so (th_probe.hip: the level code's device math functions evaluated on arrays) with the
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
    src = os.path.join(HERE, "th_probe.hip")
    deps = [src] + [os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith((".h", ".inc"))]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call([B.HIPCC] + B.FLAGS + ["-I" + B.CSRC, "-shared", src, "-o", LIB])
    return LIB


def lib():
    L = ctypes.CDLL(build())
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.icar_probe_math.argtypes = [ci, ci, vp, vp, vp]
    L.icar_probe_dec_index.argtypes = [vp, vp, ci, ci, ci, vp]
    return L


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
