import sys, types, numpy as np
sys.path.insert(0, ".")
import torch, bench
args = types.SimpleNamespace(nx=256, ny=256, nz=40, hill=1000.0, adv="mpdata", mp="thompson")
d, opt, case, g = bench.build_tile(args, 0, 1, 0)
names = ["water_vapor", "cloud_water_mass", "rain_mass", "snow_mass", "potential_temperature", "cloud_ice_mass", "graupel_mass", "cloud_ice_number", "rain_number"]
for it in range(400):
    bench.one_step(d, opt)
    if it % 100 == 99:
        st = {n: d.get(n) for n in names}
        bad = [n for n, a in st.items() if not np.isfinite(a).all()]
        neg = [n for n, a in st.items() if n != "potential_temperature" and a.min() < 0]
        print(it + 1, "non-finite:", bad, "negative:", neg, "qv max %.4f th range %.1f..%.1f precip max %.3f" % (st["water_vapor"].max(), st["potential_temperature"].min(), st["potential_temperature"].max(), d.get("accumulated_precipitation").max()), flush=True)
