"""
How many host cores this process may use.

A GPU box hands each GPU process a share of the host (16 cores per GPU on the MI355X pool) while `os.cpu_count()` reports
every core of the machine: worker pools sized by `os.cpu_count()` oversubscribe the share by an order of magnitude (a fork
pool of 256 NumPy workers ran 50x slower per task than 16).  Order of precedence: ALIBY_HOST_CORES, the cgroup CPU quota,
the affinity mask capped at 16 per rank; quota and mask are divided by LOCAL_WORLD_SIZE (the ranks of a node share them).
"""

from __future__ import annotations

import os


def _cgroup_quota() -> float | None:
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cores(default_cap: int = 16) -> int:
    env = os.environ.get("ALIBY_HOST_CORES")
    if env:
        return max(1, int(env))
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # ranks of one node (torch.distributed.run sets LOCAL_WORLD_SIZE) share the node's quota / affinity mask
    try:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    except ValueError:
        local = 1
    quota = _cgroup_quota()
    if quota is not None:
        return max(1, min(affinity, int(quota + 0.5)) // local)
    return max(1, min(affinity // local, default_cap))


def describe() -> dict:
    return {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota": _cgroup_quota(), "usable": usable_cores()}


def cpu_stat() -> dict:
    """The cgroup's CPU accounting (usage_usec, nr_periods, nr_throttled, throttled_usec ...), {} where there is none: a process
    group that runs into its quota is stopped for the rest of the period — every thread, the launch thread included."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            return {k: int(v) for k, v in (line.split()[:2] for line in open(path).read().splitlines() if line.strip())}
        except (OSError, ValueError):
            continue
    return {}
