"""python profiles/collect_resources.py [round] -> profiles/rNN_resources.json: registers / scratch / spills / occupancy of the
kernels the step spends its time in, as compiled by the product's flags (icar_amd.build.kernel_resources)."""
import json, subprocess, sys
sys.path.insert(0, ".")
from icar_amd import build as B
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
want = {"mpdata.hip": ["k_mpdata_fused<5, true, true, true>", "k_mpdata_fused<5, true, true, false>", "k_mpdata_coef<false>"],
        "mp_thompson.hip": ["k_thompson_pack<512>", "k_thompson_pack<1024>"], "capi.hip": ["k_max_courant"], "advect.hip": None, "step.hip": None}
out = {"hipcc": subprocess.run([B.HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0]}
for src, names in want.items():
    res = B.kernel_resources(src)
    out[src] = {k: v for k, v in res.items() if names is None or k in names}
json.dump(out, open(f"profiles/{rnd}_resources.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
