import os, subprocess, sys
sys.path.insert(0, "/root/repo")
from icar_amd import build as B
tag, src = sys.argv[1], sys.argv[2]; extra = sys.argv[3:]
obj = os.path.join(B.LIBDIR, "ab", src.replace(".hip", f"_{tag}.o"))
subprocess.check_call([B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.LIBDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(B.LIBDIR, "ab", f"lib_{tag}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
r = B.kernel_resources(src, extra)
k = [v for n, v in r.items() if "k_thompson_pack<512>" in n or "k_mpdata_fused<5, true, true, true>" in n]
print(tag, lib, k)
