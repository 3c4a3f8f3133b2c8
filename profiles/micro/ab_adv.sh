#!/bin/bash
# A/B of the builds under icar_amd/lib/ab: advect alone (prof_advect.py), two runs each, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do
for so in icar_amd/lib/ab/lib_*.so; do
  n=$(basename $so .so)
  echo "$n $(ICAR_HIP_LIB=$R/$so timeout 120 python profiles/prof_advect.py ${SIZE:-512} 30 2>&1 | tail -1)"
done; done
