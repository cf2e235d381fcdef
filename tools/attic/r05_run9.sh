#!/bin/bash
echo "cgroup:"; cat /proc/self/cgroup; echo "cpu.max:"; cat /sys/fs/cgroup/cpu.max 2>&1; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1 | head -3
p=$(sed -n 's/^0:://p' /proc/self/cgroup); echo "path $p"; cat /sys/fs/cgroup$p/cpu.max 2>&1
cat /sys/fs/cgroup/cpu.stat 2>&1 | head -8
python - <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)))
PY
