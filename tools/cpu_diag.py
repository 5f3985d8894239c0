import os, glob
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ["/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/self/cgroup"]:
    try: print(f, open(f).read().strip()[:300])
    except Exception as e: print(f, "->", e)
os.system("nproc; ulimit -u; cat /proc/meminfo | head -3")
