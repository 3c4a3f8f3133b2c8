#!/bin/bash
# per-kernel average durations of a bench run in MPDATA's exact mode (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
O=gpurun_out/ks_exact; rm -rf $O; mkdir -p $O
timeout 240 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-later-window --mpdata-exact > $O/trace.log 2>&1
db=$(find $O/trace -name '*.db' | head -1)
python profiles/summarize_rocpd.py $db | head -14
