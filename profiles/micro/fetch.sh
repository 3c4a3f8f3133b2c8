#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per MPDATA kernel for one bench configuration: usage fetch.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/fs_$tag; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- $P > $O/f.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- $P > $O/w.log 2>&1
echo "== $tag $@"
python - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("f", "w"):
    for r in csv.DictReader(open(f"{O}/{d}/p_counter_collection.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_upwind_pass", "k_mpdata_fluxes_pipe", "k_mpdata_fluxes", "k_mpdata_final2", "k_thompson_pack"):
    if k in acc:
        f = acc[k]["FETCH_SIZE"]; w = acc[k]["WRITE_SIZE"]
        print(f"{k:24s} read {2*1024*sum(f)/len(f)/1e6:8.1f} MB   write {1024*sum(w)/max(1,len(w))/1e6:8.1f} MB")
PY
