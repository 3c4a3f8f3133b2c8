set -x
export TMPDIR=/tmp
python profiles/micro/dbg_config3_theta.py prepare
for t in 0010e3c 33e8be1 7c853ab nocontract; do ICAR_HIP_LIB=$PWD/icar_amd/lib/libicar_hip_$t.so timeout 300 python profiles/micro/dbg_config3_theta.py run $t 2>&1 | grep -v "^+" ; done
timeout 300 python profiles/micro/dbg_config3_theta.py run head
