#!/bin/bash
# advect alone at two column heights for the builds under icar_amd/lib/ab (waves per block experiment)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do for so in icar_amd/lib/ab/lib_*.so; do n=$(basename $so .so)
  for nz in 40 36 24; do echo "$n $(ICAR_HIP_LIB=$R/$so timeout 120 python profiles/prof_advect.py 512 20 $nz 2>&1 | tail -1)"; done
done; done
