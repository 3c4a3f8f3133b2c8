export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_advect.py -x -q 2>&1 | tail -3
python profiles/micro/ab.py -n 6 --tag rv gvreload=icar_amd/lib/ab/lib_gvreload.so rvlds=icar_amd/lib/libicar_hip.so 2>&1 | tail -3
