#!/usr/bin/env python3
"""tools/numa_probe.py -- is the occasional slow PCIe-inclusive pass of a short search (bench.py --workload q1 --nseq 100000: 7.4 instead of
3.0 ms in five runs of eight, page-locked host buffers only) a matter of which NUMA node the process runs on?  Runs bench.py with the
process pinned to the CPUs of each node in turn and prints the GPU's own node."""
import glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", [(os.path.basename(n), open(n + "/cpulist").read().strip()) for n in nodes])
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip(), open(os.path.dirname(f) + "/uevent").read().split("PCI_SLOT_NAME=")[-1].split()[0])
def cpus(n):
    out = []
    for part in open(n + "/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
for n in nodes + [None]:
    for rep in range(3):
        pre = (lambda c=cpus(n): os.sched_setaffinity(0, c)) if n else None
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "q1", "--nseq", "100000", "--steps", "10", "--warmup", "3", "--cpu-seconds", "0"],
                           capture_output=True, text=True, preexec_fn=pre)
        d = json.loads([l for l in p.stdout.split("\n") if l.startswith("{")][-1])
        print(os.path.basename(n) if n else "unpinned", rep, "GCUPS", d["value"], "inclusive pinned ms", d["pcie_inclusive"]["ms"], "pageable GCUPS", d["pcie_inclusive_pageable"], flush=True)
