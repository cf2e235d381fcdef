#!/usr/bin/env python3
"""tools/collect_final.py ROUND -- copy what tools/final_validation.sh left under gpurun_out/final/ (scratch) into profiles/
(tracked): the bench lines, the shard-balance prediction, the CLI phases, the microbenchmark, the rocprofv3 summaries."""
import json, os, shutil, subprocess, sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "final"), os.path.join(root, "profiles")
for f in sorted(os.listdir(src)):
    if f.startswith("bench_") and f.endswith(".json"):
        lines = [l for l in open(os.path.join(src, f)).read().split("\n") if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"r{rnd}_{f}"), "w") as o:
                json.dump(json.loads(lines[-1]), o, indent=1)
for a, b in (("shard_balance.txt", f"r{rnd}_shard_balance_1gpu.txt"), ("cli_1m.txt", f"r{rnd}_cli_1m_phases.txt"), ("oprate_q8.txt", f"r{rnd}_oprate_q8.txt"),
             ("oprate4.txt", f"r{rnd}_oprate4_valu_issue.txt"), ("oprate5.txt", f"r{rnd}_oprate5_pause_placement.txt"), ("oprate6.txt", f"r{rnd}_oprate6_lds_pairing.txt"), ("oprate7.txt", f"r{rnd}_oprate7_sdwa_bytemax.txt"), ("oprate8.txt", f"r{rnd}_oprate8_pk_mad.txt"), ("oprate9.txt", f"r{rnd}_oprate9_int32_row.txt"), ("q1_tail.txt", f"r{rnd}_q1_c5_tail.txt"), ("startup.txt", f"r{rnd}_cli_startup.txt"),
             ("cli_q1_1m.txt", f"r{rnd}_cli_q1_1m_phases.txt"), ("cli_hybrid.txt", f"r{rnd}_cli_hybrid.txt"), ("inclusive_probe_q1.txt", f"r{rnd}_inclusive_probe_q1.txt"),
             ("inclusive_probe_c4.txt", f"r{rnd}_inclusive_probe_c4.txt"), ("pin_probe.txt", f"r{rnd}_pin_probe.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
with open(os.path.join(dst, f"r{rnd}_pytest_gpu_tail.txt"), "w") as o:
    o.write("".join(open(os.path.join(src, "pytest_gpu.log")).readlines()[-3:]))
for wl, nseq, label in (("c2", "1000000", "c4_1gpu"), ("c2", "100000", "c2"), ("c3", "100000", "c3_int8"), ("c5", "100000", "c5"), ("q1", "100000", "q1_100k"), ("q1", "1000000", "q1_1m")):
    rc = subprocess.call([sys.executable, os.path.join(root, "tools", "collect_prof.py"), rnd, wl, nseq, label])
    if rc:
        print(f"collect_final: the rocprofv3 summary of {wl} {nseq} was NOT collected (rc {rc})")
