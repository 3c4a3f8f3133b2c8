"""A/B builds of one source with macro sets: python profiles/micro/build_ab.py <file.hip> name1:-DA,-DB name2: ...
-> icar_amd/lib/ab/lib_<name>.so (other objects reused).  A name of the form name@path compiles <path> instead of csrc/<file.hip>."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, ".")
from icar_amd import build as B
src = sys.argv[1]
B.build()
os.makedirs(os.path.join(B.LIBDIR, "ab"), exist_ok=True)
def one(spec):
    name, flags = spec.split(":", 1)
    path = os.path.join(B.CSRC, src)
    if "@" in name:
        name, path = name.split("@", 1)
    extra = [f for f in flags.split(",") if f]
    obj = os.path.join(B.LIBDIR, "ab", f"{name}.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + extra + ["-I" + B.CSRC, "-c", path, "-o", obj])
    objs = [obj if s == src else os.path.join(B.LIBDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
    lib = os.path.join(B.LIBDIR, "ab", f"lib_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return lib
with ThreadPoolExecutor(6) as ex:
    for l in ex.map(one, sys.argv[2:]):
        print(l)
