export TMPDIR=/tmp
python profiles/micro/ab.py -n 5 --tag diag exact=icar_amd/lib/libicar_hip.so noloads_approx=icar_amd/lib/ab/lib_diagA.so loads_approx=icar_amd/lib/ab/lib_diagB.so 2>&1 | tail -5
