#!/bin/bash
# profiles/run_profiles.sh -- run on the GPU box (gpurun): bench line, rocprofv3 kernel trace of the same command, and
# three PMC passes (SQ counters / FETCH_SIZE / WRITE_SIZE -- separate passes, no trace domains mixed in, as
# MI355X_MICROARCH.md prescribes).  Outputs land in gpurun_out/<round>/ (default r02), post-processed by
# profiles/collect_round.py <round>.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
RND=${1:-r04}
O=gpurun_out/$RND; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err      # the driver's command
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $B > $O/trace.log 2>&1
P="python bench.py --steps 4 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- $P > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $P > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $P > $O/pmc_write.log 2>&1
timeout 300 python profiles/prof_winds.py > $O/winds.json 2> $O/winds.err
ls -R $O | head -40
tail -c 600 $O/bench_line.json
