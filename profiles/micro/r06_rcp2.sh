export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_advect.py tests/test_gpu_trajectory.py -x -q 2>&1 | tail -4
python profiles/micro/ab.py -n 6 --tag rcp2 loads=icar_amd/lib/ab/lib_prev.so newton=icar_amd/lib/libicar_hip.so 2>&1 | tail -3
