#!/bin/bash
# arbitrary SQ counter pass per kernel: usage pmc_any.sh <tag> "<counters>" [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; ctr=$2; shift; shift
O=gpurun_out/pa_$tag; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
echo "== $tag $ctr"
python - $O <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1] + "/s/p_counter_collection.csv")):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_thompson_pack", "k_mpdata_fused", "k_mp_simple_pack", "k_w6_fall_tile", "k_w6_rates", "k_w6_prep", "k_wsm3_fall_tile", "k_wsm3_rates", "k_wsm3_prep"):
    if k in acc: print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())})
PY
