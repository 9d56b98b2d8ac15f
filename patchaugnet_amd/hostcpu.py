"""Host CPU budget of this process.

On the GPU boxes ``os.cpu_count()`` reports every core of the machine (256) while the container's cgroup grants a CPU
quota of a few cores (cpu.max "1600000 100000" = 16).  A CPU torch op or an OpenMP region sized by ``cpu_count`` then spins
hundreds of threads, burns the quota and the kernel throttles the WHOLE process for the rest of the 100 ms period: 35-60 ms
stalls that show up in whichever host call happens to run (measured: a config-4 training step goes 21 ms -> 47 ms average).
``cpu_budget()`` is the number of threads that fit the grant; ``limit_host_threads()`` applies it to ATen and OpenMP.
"""
import math
import os


def _cgroup_quota():
    try:                                              # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:                                              # cgroup v1
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def cpu_budget():
    """Threads this process can run without being throttled: min(affinity mask, cgroup quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = _cgroup_quota()
    if quota is not None:
        n = min(n, max(1, math.floor(quota)))
    return max(1, n)


def limit_host_threads(n=None):
    """Cap ATen's intra-op pool and OpenMP at the CPU budget (or ``n``).  Returns the thread count applied."""
    import torch
    n = int(n or cpu_budget())
    os.environ["OMP_NUM_THREADS"] = str(n)
    torch.set_num_threads(n)
    return n
