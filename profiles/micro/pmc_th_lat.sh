#!/bin/bash
# memory-latency counters of the Thompson kernel: usage pmc_th_lat.sh <tag> <nx> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; nx=$2; shift 2
O=gpurun_out/thl_$tag; rm -rf $O; mkdir -p $O
P="python profiles/prof_thompson.py $nx"
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $O/t -o p -- $P > $O/t.log 2>&1
echo "== $tag $@"
python profiles/summarize_pmc.py $O/summary.md $O/s/p_counter_collection.csv $O/t/p_counter_collection.csv | grep -E "kernel|k_thompson_[pm]" | cut -c1-600
