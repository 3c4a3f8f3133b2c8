#!/bin/bash
# per-kernel average durations of one bench run (rocprofv3 --kernel-trace): usage kstats.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/ks_$tag; rm -rf $O; mkdir -p $O
timeout 240 env "$@" rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace.log 2>&1
db=$(find $O/trace -name '*.db' | head -1)
echo "== $tag $@"
python profiles/summarize_rocpd.py $db | grep -E "k_upwind_pass|k_mpdata_fluxes|k_mpdata_final2|k_thompson_pack" | awk -F'|' '{print $2, $5}'
