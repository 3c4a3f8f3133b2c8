#!/bin/bash
# profiles/run_configs.sh -- one bench line per BASELINE.json configuration that fits one MI355X (the multi-GPU configurations
# as the tile one GPU would own).  Output: gpurun_out/configs.jsonl (copied to profiles/<round>_configs.jsonl).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
O=gpurun_out/configs.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 >> $O; }
run --nx 100 --ny 100 --nz 30 --adv upwind --mp simple      # configs[0]
run --nx 256 --ny 256 --nz 40                               # configs[1]
run --nx 512 --ny 512 --nz 40                               # configs[2] = the headline grid on one GPU
run --nx 258 --ny 512 --nz 40                               # its tile on 2 GPUs (2x1, strong scaling)
run --nx 258 --ny 258 --nz 40                               # its tile on 4 GPUs (2x2)
run --nx 258 --ny 130 --nz 40                               # its tile on 8 GPUs (2x4)
run --nx 512 --ny 256 --nz 40                               # configs[3] tile of 1024x1024 on 2x4 images
run --nx 256 --ny 128 --nz 80                               # configs[4] tile of 512x512x80 on 2x4 images (microphysics + advection part)
run --nx 1024 --ny 1024 --nz 40                             # configs[3] whole domain on one GPU
run --mp wsm3; run --mp wsm6; run --mp simple               # the other microphysics slots on the headline grid
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:60], "| ms/step", round(d["ms_per_step"], 3), "| cells/s %.3e" % d["value"])
PY
