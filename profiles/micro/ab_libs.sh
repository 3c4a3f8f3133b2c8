#!/bin/bash
# profiles/micro/ab_libs.sh -- A/B of library builds under icar_amd/lib/ab/lib_<name>.so (ICAR_HIP_LIB override): bench line per build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for so in icar_amd/lib/ab/lib_*.so; do
  n=$(basename $so .so)
  for rep in 1 2; do
    ICAR_HIP_LIB=$R/$so timeout 200 python bench.py --no-cpu-baseline --steps ${STEPS:-40} --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', 'ms/step %.3f' % d['ms_per_step'], 'advect kernel ms %.4f' % (r.get('avg_ms') or 0), 'mp ms/step %.4f' % (r.get('mp_ms_per_step') or 0))"
  done
done
