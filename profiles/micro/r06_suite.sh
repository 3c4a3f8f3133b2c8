export TMPDIR=/tmp
rm -rf gpurun_out/parity
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -5
