#!/bin/bash
# per-kernel durations for a bench configuration: usage kstats_mp.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/km_$tag; rm -rf $O; mkdir -p $O
timeout 240 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/trace.log 2>&1
db=$(find $O/trace -name '*.db' | head -1)
python profiles/summarize_rocpd.py $db | head -12 | cut -c1-140
