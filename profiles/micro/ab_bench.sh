#!/bin/bash
# the bench step with every build under icar_amd/lib/ab, two runs each: ab_bench.sh [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do
for so in icar_amd/lib/ab/lib_*.so; do
  n=$(basename $so .so)
  ICAR_HIP_LIB=$R/$so timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', 'ms/step %.4f' % d['ms_per_step'], 'advect %.4f' % (r.get('avg_ms') or 0), 'mp %.4f' % (r.get('mp_ms_per_step') or 0))"
done; done
