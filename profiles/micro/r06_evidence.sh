export TMPDIR=/tmp
rm -rf gpurun_out/parity
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8
bash profiles/run_profiles.sh r06 2>&1 | tail -5
bash profiles/run_configs.sh 2>&1 | tail -16
