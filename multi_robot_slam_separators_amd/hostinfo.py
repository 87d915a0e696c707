"""Host CPU share of this process: the cgroup CPU quota when there is one, else the affinity mask.

A GPU box hands a job a FRACTION of the host (e.g. 16 CPUs of 256 through cgroup cpu.max); thread pools
sized from os.cpu_count() then oversubscribe the quota and the kernel's CFS bandwidth control stalls the
whole process for tens of milliseconds at a time (seen as periodic 40-80 ms gaps in bench steps)."""
import math
import os


def cpu_share():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                             # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        try:                                         # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except Exception:
            pass
    return max(1, n)
