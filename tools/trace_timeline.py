#!/usr/bin/env python3
"""tools/trace_timeline.py DIR -- merge rocprofv3's kernel trace and memory-copy trace of one run into one timeline (ms from the
first event), kernels shorter than 20 us and copies smaller than 64 KB summarised per gap."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:28] + f" grid {r.get('Grid_Size_X','')}"))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r.get('Direction','')} {int(r.get('Size', r.get('Bytes', 0)) or 0)/1e6:.2f} MB"))
ev.sort()
t0 = ev[0][0]
for s, e, n in ev:
    if e - s < 20000 and not n.startswith("K osw_sw"):
        continue
    print(f"{(s-t0)/1e6:10.3f} -> {(e-t0)/1e6:10.3f}  ({(e-s)/1e6:8.3f} ms)  {n}")
