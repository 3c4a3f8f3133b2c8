# read-only probe: can this lease expose more than one logical device (CPX / DPX compute partitions)?  
rocm-smi --showcomputepartition 2>&1 | tail -8
rocm-smi --showmemorypartition 2>&1 | tail -6
for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition; do echo "$f: $(cat $f 2>&1)"; done
ls -la /sys/class/drm/card*/device/current_compute_partition 2>&1
python -c "import torch; print('devices', torch.cuda.device_count())"
rocminfo 2>/dev/null | grep -c "gfx950"
id -u; cat /proc/1/cgroup 2>/dev/null | head -3; ls /dev/dri /dev/kfd 2>&1 | head
