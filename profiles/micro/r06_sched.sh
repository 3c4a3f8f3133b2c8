export TMPDIR=/tmp
python profiles/micro/ab.py -n 5 --tag sched default=icar_amd/lib/libicar_hip.so maxilp=icar_amd/lib/ab/lib_maxilp.so itermin=icar_amd/lib/ab/lib_itermin.so 2>&1 | tail -5
