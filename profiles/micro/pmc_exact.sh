#!/bin/bash
# PMC passes over the exact-mode MPDATA kernels (bench.py --mpdata-exact): SQ counters, FETCH_SIZE, WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
O=gpurun_out/pmc_exact; rm -rf $O; mkdir -p $O
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-later-window --no-kernel-timers --mpdata-exact"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -o p -- $P > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $P > $O/write.log 2>&1
python profiles/summarize_pmc.py $O/summary.md $O/sq/p_counter_collection.csv $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv > /dev/null
grep -E "kernel|---|k_mpx|k_upwind" $O/summary.md
