export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
python profiles/micro/ab.py -n 5 --tag sync formal=icar_amd/lib/libicar_hip.so relaxed=icar_amd/lib/libicar_hip_relaxed.so 2>&1 | tail -4
