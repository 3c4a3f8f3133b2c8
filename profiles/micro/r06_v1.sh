export TMPDIR=/tmp
python profiles/micro/ab.py -n 6 --tag v1 default=icar_amd/lib/libicar_hip.so v1=icar_amd/lib/ab/lib_v1.so 2>&1 | tail -4
