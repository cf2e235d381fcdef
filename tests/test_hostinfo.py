"""The hardware threads a process may really keep busy: affinity mask AND cgroup CPU bandwidth (oswald::usable_cpus in the C++
host library, oswald_amd/hostinfo.py for bench.py).  A container with 16 CPUs' worth of a 256-thread host shows 256 threads; a
team of 128 is throttled for most of every scheduling period (round 5: the hybrid mode's accelerator test)."""
import ctypes as C
import os

import hostlib
from oswald_amd import hostinfo


def _both(root):
    lib = hostlib.load()
    lib.oswald_host_usable_cpus.restype = C.c_uint
    lib.oswald_host_usable_cpus.argtypes = [C.c_char_p]
    a, b = hostinfo.usable_cpus(str(root)), lib.oswald_host_usable_cpus(str(root).encode())
    assert a == b
    return a


def test_cgroup_quota_caps_the_team(tmp_path):
    have = len(os.sched_getaffinity(0))
    assert _both(tmp_path) == have                                   # no cgroup files: the affinity mask
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert _both(tmp_path) == have                                   # cgroup v2, no limit
    (tmp_path / "cpu.max").write_text("250000 100000\n")
    assert _both(tmp_path) == min(have, 2)                           # 2.5 CPUs' worth: two threads
    (tmp_path / "cpu.max").write_text("50000 100000\n")
    assert _both(tmp_path) == 1                                      # less than one: one
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert _both(tmp_path) == min(have, 16)                          # the round-5 GPU box
    (tmp_path / "cpu.max").unlink()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")   # cgroup v1
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert _both(tmp_path) == min(have, 3)
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert _both(tmp_path) == have
