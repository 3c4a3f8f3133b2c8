#!/usr/bin/env python
"""Post-process gpurun_out/<round> (profiles/run_profiles.sh <round>) into the committed evidence of that round (default r02):
  profiles/<round>_bench_line.json      the bench.py JSON line
  profiles/<round>_kernel_stats.md      rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3`
  profiles/<round>_pmc.md               PMC passes (SQ / FETCH_SIZE / WRITE_SIZE) per kernel, VALU busy, HBM traffic
  profiles/advect_traffic.json      HBM bytes per advect() call (bench.py roofline.traffic)
  profiles/<round>_winds.json           LUT build / spatial_winds timing"""
import csv, glob, io, json, os, shutil, subprocess, sys
from collections import defaultdict
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
O = os.path.join(ROOT, "gpurun_out", RND)
shutil.copy(os.path.join(O, "bench_line.json"), os.path.join(HERE, RND + "_bench_line.json"))
if os.path.exists(os.path.join(O, "winds.json")):
    shutil.copy(os.path.join(O, "winds.json"), os.path.join(HERE, RND + "_winds.json"))
db = max(glob.glob(os.path.join(O, "trace", "**", "*.db"), recursive=True), key=os.path.getmtime)   # gpurun merges: older runs may linger
md = os.path.join(HERE, RND + "_kernel_stats.md")
open(md, "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline (MI355X, 512x512x40, N=1)\n\n")
subprocess.check_call([sys.executable, os.path.join(HERE, "summarize_rocpd.py"), db, md], stdout=subprocess.DEVNULL)
pm = os.path.join(HERE, RND + "_pmc.md")
open(pm, "w").write("# rocprofv3 --pmc passes over `python bench.py --steps 4 --warmup 2 --no-cpu-baseline` (three separate passes)\n\n"
                    + open(os.path.join(HERE, "summarize_pmc.py")).read().split('"""')[1].split("usage:")[0].strip() + "\n\n")
files = [os.path.join(O, d, "p_counter_collection.csv") for d in ("pmc_sq", "pmc_fetch", "pmc_write")]
subprocess.check_call([sys.executable, os.path.join(HERE, "summarize_pmc.py"), pm] + files, stdout=subprocess.DEVNULL)
# HBM bytes per advect() call: sum over its kernels of 2*FETCH_SIZE + WRITE_SIZE (KB), one dispatch of each per call
per = defaultdict(lambda: defaultdict(list))
for f in files[1:]:
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
adv = [k for k in per if any(t in k for t in ("k_upwind_pass", "k_mpdata_fused"))]
rd = sum(2 * 1024 * sum(per[k]["FETCH_SIZE"]) / len(per[k]["FETCH_SIZE"]) for k in adv)
wr = sum(1024 * sum(per[k]["WRITE_SIZE"]) / len(per[k]["WRITE_SIZE"]) for k in adv)
line = json.loads(open(os.path.join(O, "bench_line.json")).read().strip().splitlines()[-1])
tm = line["config"]["tile_memory"]
json.dump({"hbm_bytes_per_advect_call": rd + wr, "read_bytes": rd, "write_bytes": wr, "kernels": sorted(adv), "round": RND,
           # bench.py attaches this figure to its roofline only for exactly this configuration and kernel generation
           "config": {"nx": tm[0], "ny": tm[2], "nz": tm[1], "adv": "mpdata", "nscalars": 9, "kernels": line["roofline"]["kernel"].split("(", 1)[1].rstrip(")"),
                      "generation": __import__("re").search(r'KERNEL_GENERATION = "([^"]*)"', open(os.path.join(ROOT, "bench.py")).read()).group(1)},
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, "
                     "MI355X_MICROARCH.md); mean per dispatch, one dispatch of each kernel per advect() call; 512x512x40, 9 scalars"},
          open(os.path.join(HERE, "advect_traffic.json"), "w"), indent=1)
print(open(md).read()); print(json.load(open(os.path.join(HERE, "advect_traffic.json"))))
