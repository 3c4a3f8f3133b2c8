#!/bin/bash
# PC sampling (rocprofv3, stochastic) of one driver script with a -gline-tables-only build of the library:
#   pcs.sh <tag> <lib.so> <python script + args...>
# The raw CSV stays on the box; profiles/micro/pcs_agg.py leaves per-line / per-opcode / per-stall-reason counts in gpurun_out/pcs_<tag>.md
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; lib=$2; shift 2
O=/tmp/pcs_$tag; rm -rf $O; mkdir -p $O gpurun_out
export ICAR_HIP_LIB=$R/$lib
rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60
for method in stochastic host_trap; do
  unit=cycles; iv=${PCS_IV:-65536}
  if [ $method = host_trap ]; then unit=time; iv=100; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
      --kernel-trace --output-format csv -d $O/$method -o p -- "$@" > gpurun_out/pcs_${tag}_$method.log 2>&1
  echo "rc $? ($method)"; tail -3 gpurun_out/pcs_${tag}_$method.log
  if ls $O/$method/*pc_sampling*.csv $O/$method/*/*pc_sampling*.csv > /dev/null 2>&1; then break; fi
done
find $O -type f | head -20
python profiles/micro/pcs_agg.py $O gpurun_out/pcs_$tag.md | head -80
