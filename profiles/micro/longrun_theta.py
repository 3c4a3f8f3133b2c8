import sys, types, numpy as np
sys.path.insert(0, ".")
import torch, bench
args = types.SimpleNamespace(nx=256, ny=256, nz=40, hill=1000.0, adv="mpdata", mp="thompson")
d, opt, case, g = bench.build_tile(args, 0, 1, 0)
th0 = d.get("potential_temperature").copy()
for it in range(400):
    bench.one_step(d, opt)
    if it % 100 == 99:
        th = d.get("potential_temperature")
        jmx, kmx, imx = np.unravel_index(th.argmax(), th.shape); jmn, kmn, imn = np.unravel_index(th.argmin(), th.shape)
        print(it + 1, "th max %.1f at (j,k,i)=(%d,%d,%d) [was %.1f]; min %.1f at (%d,%d,%d) [was %.1f]" % (th.max(), jmx, kmx, imx, th0[jmx, kmx, imx], th.min(), jmn, kmn, imn, th0[jmn, kmn, imn]), flush=True)
w = d.get("w"); print("w range", w.min(), w.max(), "w top level range", w[:, -1].min(), w[:, -1].max())
