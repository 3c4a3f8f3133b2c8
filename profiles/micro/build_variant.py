"""A/B builds for kernel experiments: python profiles/micro/build_variant.py <tag> <file.hip> [-DFLAG ...]
-> icar_amd/lib/libicar_hip_<tag>.so with <file.hip> recompiled with the extra flags (the other objects are reused).
Select it at run time with ICAR_HIP_LIB=icar_amd/lib/libicar_hip_<tag>.so."""
import os, subprocess, sys
sys.path.insert(0, ".")
from icar_amd import build as B
tag, src = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
B.build()
obj = os.path.join(B.LIBDIR, src.replace(".hip", f"_{tag}.o"))
subprocess.check_call([B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.LIBDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(B.LIBDIR, f"libicar_hip_{tag}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
print(lib)
