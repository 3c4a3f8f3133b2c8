"""debug: evaluate a traced list of the oracle's libm calls (oracle/thompson_column.c: orc_thompson_trace) with the device's functions"""
import sys, ctypes, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from icar_amd import ideal
from icar_amd.capi import lib, check
from util import single_image_domain
tr = np.load(sys.argv[1])
c = ideal.make_case(8, 8, 4)
d = single_image_domain(c)
for op in (0, 1, 2, 3, 4, 6):
    sel = tr[tr[:, 0] == op]
    if not len(sel): continue
    x = np.ascontiguousarray(sel[:, 1]); y = np.ascontiguousarray(sel[:, 2]); out = np.zeros(len(sel))
    check(lib().icar_hip_thompson_math_probe(d.ctx, op, len(sel), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)), "probe")
    want = sel[:, 3]
    bad = np.nonzero(out.view(np.int64) != want.view(np.int64))[0]
    print("op", op, "calls", len(sel), "differ", len(bad))
    for b in bad[:10]:
        print("   x=%r y=%r dev=%r host=%r" % (float(x[b]), float(y[b]), float(out[b]), float(want[b])), " bits", hex(out.view(np.int64)[b]), hex(want.view(np.int64)[b]))
