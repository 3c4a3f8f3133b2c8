#!/bin/bash
# bench.py with the step loop's graph replays on / off (timers off: they force the eager loop), per tile size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for sz in "512 512" "258 258" "258 130"; do set -- $sz
  for g in 1 0; do for r in 1 2; do
    ICAR_BENCH_GRAPH=$g python bench.py --no-cpu-baseline --no-later-window --no-kernel-timers --steps 40 --warmup 6 --nx $1 --ny $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2 graph=$g ms/step %.4f' % d['ms_per_step'], 'replays', d['config'].get('graph_replays'))"
  done; done; done
