import os
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print(open("/proc/loadavg").read())
