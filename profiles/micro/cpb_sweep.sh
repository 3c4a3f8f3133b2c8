#!/bin/bash
# Thompson layout sweep: columns per block (ICAR_HIP_THOMPSON_CPB) -> ms per step
for c in "$@"; do
  ICAR_HIP_THOMPSON_CPB=$c timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpb', $c, round(d['ms_per_step'],3), round(d['microphysics']['ms_per_step'],3))"
done
