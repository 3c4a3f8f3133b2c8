"""the regime cases of tests/test_gpu_thompson.py (warm mixed phase, cold graupel, long dt) at a size where one-in-1e7 events show:
Thompson alone with cooling between calls, device against oracle, counts of cells that differ in any bit"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_thompson as T
from oracle import orc
from icar_amd.options import options_t
orc.build(); p_, f_ = options_t().mp_options.as_arrays(); orc.thompson_init(p_, f_)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for name, cs in T.CASES.items():
    cs = dict(cs); cs["nx"] = cs["ny"] = n; cs["nz"] = 40
    out, ref = T.run_case(orc, mode=0, **cs)
    tot = 0
    for k in out:
        a, b = np.ascontiguousarray(out[k]), np.ascontiguousarray(ref[k])
        nd = int((a.view(np.int64 if a.dtype == np.float64 else np.int32) != b.astype(a.dtype).view(np.int64 if a.dtype == np.float64 else np.int32)).sum())
        tot += nd
        if nd: print("  ", name, k, nd, "of", a.size, "differ")
    print(name, cs["steps"], "calls at", n, "x", n, "x 40:", "bit-identical" if tot == 0 else f"{tot} values differ", flush=True)
