export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_advect.py -x -q 2>&1 | tail -6
python profiles/micro/ab.py -n 5 --tag plan base=icar_amd/lib/ab/lib_base.so plan=icar_amd/lib/libicar_hip.so 2>&1 | tail -4
python profiles/micro/ab.py -n 5 --tag plan_tile --bench-args "--nx 258 --ny 130" base=icar_amd/lib/ab/lib_base.so plan=icar_amd/lib/libicar_hip.so 2>&1 | tail -4
