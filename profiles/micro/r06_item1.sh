set -x
export TMPDIR=/tmp
python profiles/micro/dbg_config3_theta.py prepare
timeout 300 python profiles/micro/dbg_config3_theta.py run new
timeout 1500 python -m pytest tests/test_gpu_advect.py -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 2>&1 | tail -3
