#!/usr/bin/env python3
"""tools/collect_prof.py ROUND WORKLOAD NSEQ [LABEL [PROF_LABEL]] -- copy what tools/profile_gpu.sh left under
gpurun_out/prof_<workload>_<nseq>[_<prof_label>]/ (scratch) into profiles/ (tracked): r<ROUND>_<label>_summary.json,
..._kernel_stats.csv, ..._domain_stats.csv and traffic_<workload>_<nseq>.json (the file bench.py reads for
roofline.traffic).  Refuses a summary whose kernel trace and bench line do not describe the same run (the trace's span of
a chunk search must be within 3 % of the kernel time bench.py measured with HIP events in that run)."""
import glob, json, os, shutil, sys

rnd, wl, nseq = sys.argv[1], sys.argv[2], sys.argv[3]
label = sys.argv[4] if len(sys.argv) > 4 else wl
plabel = sys.argv[5] if len(sys.argv) > 5 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{wl}_{nseq}" + (f"_{plabel}" if plabel else ""))
dst = os.path.join(root, "profiles")
text = open(os.path.join(src, "summary.txt")).read()
summary = json.loads(text[text.index("{"):])
c = summary.get("consistency")
if not c or not c.get("ok"):
    sys.exit(f"collect_prof: {src}: kernel trace and bench line disagree or are missing ({c}); not collected")
with open(os.path.join(dst, f"r{rnd}_{label}_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(dst, f"r{rnd}_{label}_kernel_stats.csv"))
    dom = stats[-1].replace("_kernel_stats.csv", "_domain_stats.csv")
    if os.path.exists(dom):
        shutil.copy(dom, os.path.join(dst, f"r{rnd}_{label}_domain_stats.csv"))
# the traffic file(s) of the run: one per MODE (tools/summarize_prof.py names them: ..._tails0.json for OSWALD_HIP_PAIR_TAILS=0); a run
# labelled otherwise -- other cells, other flags -- is not what a bench line of that name would report
for t in glob.glob(os.path.join(src, f"traffic_{wl}_{nseq}*.json")):
    if not plabel or os.path.basename(t) == f"traffic_{wl}_{nseq}_{plabel}.json":
        shutil.copy(t, os.path.join(dst, os.path.basename(t)))
print("collected", label, "->", dst, "| trace vs bench kernel time: %+.2f %%" % (100 * c["relative_difference"]))
