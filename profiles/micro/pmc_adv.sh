#!/bin/bash
# SQ counter passes for the advection kernels of one bench run: usage pmc_adv.sh <tag> "<counters>" [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; ctr=$2; shift; shift
O=gpurun_out/pa_$tag; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
echo "== $tag $ctr"
python - $O <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + "/s/p_counter_collection.csv")):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
for k in acc:
    if k.startswith("k_mpdata") or k.startswith("k_upwind") or k.startswith("k_thompson"):
        print(k, f"us={sum(dur[k])/len(dur[k]):.1f}", {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())})
PY
