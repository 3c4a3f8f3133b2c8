export TMPDIR=/tmp
python profiles/micro/ab.py -n 6 --tag qx base=icar_amd/lib/libicar_hip.so qx_diag=icar_amd/lib/ab/lib_qx.so 2>&1 | tail -3
