export TMPDIR=/tmp
python profiles/micro/ab.py -n 5 --tag sleep default=icar_amd/lib/libicar_hip.so nosleep=icar_amd/lib/ab/lib_nosleep.so sleep4=icar_amd/lib/ab/lib_sleep4.so 2>&1 | tail -5
