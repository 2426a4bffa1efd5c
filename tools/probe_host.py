"""Host CPU facts of the GPU box (cores visible / usable) -- for sizing bench.py's cpu_baseline threads."""
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
import torch
print("torch threads", torch.get_num_threads())
