#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) + SQ instruction counters of the MPDATA kernels for the default bench configuration
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=${1:-x}; shift
O=gpurun_out/fm_$tag; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- $P > $O/f.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- $P > $O/w.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --stats -d $O/t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/t.log 2>&1
echo "== $tag $@"
python - "$O" <<'PY'
import csv, sys, collections, glob
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("f", "w", "s"):
    for fn in glob.glob(f"{O}/{d}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_mpdata_fused", "k_mpdata_coef", "k_setup_winds", "k_thompson_pack"):
    if k in acc:
        a = {c: sum(v) / len(v) for c, v in acc[k].items()}
        f = a.get("FETCH_SIZE", 0); w = a.get("WRITE_SIZE", 0)
        print(f"{k:18s} read {2*1024*f/1e6:8.1f} MB  write {1024*w/1e6:8.1f} MB  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(a.items()) if c not in ("FETCH_SIZE", "WRITE_SIZE")))
for fn in glob.glob(f"{O}/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if float(r["Percentage"]) > 1.0: print(f"  {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
