"""What the host offers this process: the hardware threads it may really keep busy (oswald::usable_cpus of the C++ host
library, oswald_amd/host/oswald_host.cpp, said again for bench.py: the affinity mask AND the cgroup's CPU bandwidth)."""
from __future__ import annotations

import os


def usable_cpus(cgroup_root: str = "/sys/fs/cgroup") -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    dirs = [cgroup_root]
    try:
        with open("/proc/self/cgroup") as f:
            for line in f:
                if line.startswith("0::"):
                    path = line[3:].strip()
                    while path and path != "/":
                        dirs.append(cgroup_root + path)
                        path = path.rsplit("/", 1)[0]
    except OSError:
        pass
    for d in dirs:
        try:
            with open(os.path.join(d, "cpu.max")) as f:
                quota, period = f.read().split()[:2]
            if quota != "max" and float(period) > 0 and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / float(period))))
        except (OSError, ValueError):
            pass
    try:
        with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as fq, open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as fp:
            quota, period = float(fq.read()), float(fp.read())
        if quota > 0 and period > 0:
            n = min(n, max(1, int(quota / period)))
    except (OSError, ValueError):
        pass
    return max(1, n)
