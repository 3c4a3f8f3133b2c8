#!/bin/bash
# SQ counters of the Thompson kernel alone (profiles/prof_thompson.py): usage pmc_th.sh <tag> <nx> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; nx=$2; shift 2
O=gpurun_out/th_$tag; rm -rf $O; mkdir -p $O
P="python profiles/prof_thompson.py $nx"
env "$@" $P
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/s -o p -- $P > $O/s.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d $O/t -o p -- $P > $O/t.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- $P > $O/f.log 2>&1
timeout 240 env "$@" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- $P > $O/w.log 2>&1
echo "== $tag $@"
python profiles/summarize_pmc.py $O/summary.md $O/s/p_counter_collection.csv $O/t/p_counter_collection.csv $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv | grep -E "kernel|k_thompson" | cut -c1-600
