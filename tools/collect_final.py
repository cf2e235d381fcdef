#!/usr/bin/env python3
"""tools/collect_final.py ROUND -- copy what tools/final_validation.sh left under gpurun_out/final_<part>/ (scratch) into profiles/
(tracked): the bench lines, the shard-balance prediction, the CLI phases, the microbenchmarks, the rocprofv3 summaries and the
traffic files (one per mode) bench.py reads."""
import json, os, shutil, subprocess, sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
src = os.path.join(root, "gpurun_out", "final_bench")
if os.path.isdir(src):
    for f in sorted(os.listdir(src)):
        if f.startswith("bench_") and f.endswith(".json"):
            lines = [l for l in open(os.path.join(src, f)).read().split("\n") if l.startswith("{")]
            if lines:
                with open(os.path.join(dst, f"r{rnd}_{f}"), "w") as o:
                    json.dump(json.loads(lines[-1]), o, indent=1)
    if os.path.exists(os.path.join(src, "shard_balance.txt")):
        shutil.copy(os.path.join(src, "shard_balance.txt"), os.path.join(dst, f"r{rnd}_shard_balance_1gpu.txt"))
    if os.path.exists(os.path.join(src, "pytest_gpu.log")):
        with open(os.path.join(dst, f"r{rnd}_pytest_gpu_tail.txt"), "w") as o:
            o.write("".join(open(os.path.join(src, "pytest_gpu.log")).readlines()[-3:]))
src = os.path.join(root, "gpurun_out", "final_cli")
if os.path.isdir(src):
    for a, b in (("cli_1m.txt", f"r{rnd}_cli_1m_phases.txt"), ("cli_q1_1m.txt", f"r{rnd}_cli_q1_1m_phases.txt"), ("cli_hybrid.txt", f"r{rnd}_cli_hybrid.txt"),
                 ("alloc_probe.txt", f"r{rnd}_alloc_probe.txt"), ("oprate4.txt", f"r{rnd}_oprate4_valu_issue.txt"), ("oprate8.txt", f"r{rnd}_oprate8_pk_mad.txt"),
                 ("oprate9.txt", f"r{rnd}_oprate9_int32_row.txt"), ("oprate_q8.txt", f"r{rnd}_oprate_q8.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for wl, nseq, label, plabel in (("c2", "1000000", "c4_1gpu", ""), ("c2", "1000000", "c4_1gpu_notails", "tails0"), ("c2", "100000", "c2", ""), ("c3", "100000", "c3_int8", ""),
                                ("c5", "100000", "c5", ""), ("q1", "100000", "q1_100k", ""), ("q1", "1000000", "q1_1m", "")):
    if not os.path.isdir(os.path.join(root, "gpurun_out", f"prof_{wl}_{nseq}" + (f"_{plabel}" if plabel else ""))):
        continue
    rc = subprocess.call([sys.executable, os.path.join(root, "tools", "collect_prof.py"), rnd, wl, nseq, label] + ([plabel] if plabel else []))
    if rc:
        print(f"collect_final: the rocprofv3 summary of {wl} {nseq} {plabel} was NOT collected (rc {rc})")
