#!/bin/bash
# SQ counters per kernel for one bench configuration: usage pmc_sq.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/sq_$tag; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
echo "== $tag $@"
python profiles/summarize_pmc.py /dev/null $O/s/p_counter_collection.csv | grep -E "kernel|k_thompson_pack|k_mpdata|k_upwind" | cut -c1-330
