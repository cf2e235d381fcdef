#!/usr/bin/env python3
"""tools/fill_design.py ROUND [TEMPLATE] -- put the round's measured figures into DESIGN.md: every @@NAME@@ placeholder is replaced
by the figure of that name taken from profiles/r<ROUND>_* (the files tools/collect_final.py copied there).  TEMPLATE: the text with
the placeholders (default: DESIGN.md itself, i.e. a first fill; the template is a working file of the writing session, not tracked --
a later validation run is filled in from the same template again).  Placeholders without a figure are left in place and listed."""
import json, os, re, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
template = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "DESIGN.md")
P = os.path.join(root, "profiles")


def bench(name):
    f = os.path.join(P, f"r{rnd}_bench_{name}.json")
    return json.load(open(f)) if os.path.exists(f) else None


def text(name):
    f = os.path.join(P, f"r{rnd}_{name}")
    return open(f).read() if os.path.exists(f) else None


def sp(x, nd=0):
    """11 770 / 340.7: thousands separated by a thin gap the way the documents write them"""
    s = f"{x:,.{nd}f}".replace(",", " ")
    return s


vals = {}
# ---- bench lines
rows = [("c4_1gpu", "**C4 database on one GPU (headline; 1 M sequences, 20 queries)**", "11 770"), ("c2", "C2 (100 000 sequences, 20 queries; configs[1])", "11 540"),
        ("c3_int16", "C3 PAM250, int16 cells", "11 570"), ("c3_int8", "C3 PAM250, int8 cells (configs[2])", "5 465"), ("c5", "C5 (one 5 000-row query × 100 000; configs[4])", "10 246"),
        ("c5_1m", "C5 × 1 M sequences", "10 740"), ("q1_100k", "Q1 (one 375-row query × 100 000)", "9 000"), ("q1_1m", "Q1 × 1 M sequences", "9 986"),
        ("q1_10m", "Q1 × 10 M sequences", "9 926"), ("10m", "20 queries × 10 M sequences (weak)", "11 760"), ("hi", "`hi` (escalation int16 → int32)", "10 370"),
        ("hi8", "`hi8` (escalation int8 → int16)", "5 480"), ("c2_int32", "C2, `cell_bits = 32`", "2 380"), ("c4_int32", "C4 database, `cell_bits = 32`", "—"),
        ("gloo4", "gloo rehearsal, 4 ranks on the one GPU (deal)", "11 880")]
table = []
for key, label, r4 in rows:
    d = bench(key)
    if not d:
        continue
    rf, inc = d["roofline"], d.get("inclusive") or {}
    valu = rf.get("valu", {}).get("frac")
    ref = (d.get("reference_traffic_model") or {}).get("frac")
    inc_s = "—"
    if inc.get("value"):
        inc_s = f"{sp(inc['value'])}, {inc['ms_per_step']:.2f} ms, {100 * (inc.get('vs_resident') or 0):+.1f} %"
    if key == "gloo4":  # (ranks that share the GPU: kernel times inflated by their contention, rank 0's inclusive pass is its shard's)
        table.append(f"| {label}; kernel times inflated by the ranks' contention | **{sp(d['value'])}** ({r4}) | {d['ms_per_step']:.2f} | — | — | — | — |")
        continue
    table.append(f"| {label} | **{sp(d['value'])}** ({r4}) | {d['ms_per_step']:.2f} | {rf.get('kernel_ms', 0):.2f} | {valu if valu is not None else '—'} | {inc_s} | {ref if ref is not None else '—'} |")
if table:
    vals["TABLE5"] = "\n".join(table)
d = bench("c4_1gpu")
if d:
    inc = d["inclusive"]
    vals.update(INCLC4=f"{inc['ms_per_step']:.1f}", RESC4=f"{d['ms_per_step']:.1f}", INCLC4P=f"{100 * inc['vs_resident']:.2f}".lstrip("+"), INCLC4G=sp(inc["value"]))
    if d.get("cpu_baseline"):
        vals["CPUGCUPS"] = sp(d["cpu_baseline"]["value"])
dn = bench("c4_1gpu_notails")
if d and dn:
    vals.update(C4TAILS=sp(d["value"]), C4NOTAILS=sp(dn["value"]), TAILGAIN=f"{100 * (d['value'] / dn['value'] - 1):.1f}")
d = bench("q1_1m")
if d:
    vals.update(INCLQ1=f"{d['inclusive']['ms_per_step']:.2f}", RESQ1=f"{d['ms_per_step']:.2f}")
d = bench("q1_100k")
if d:
    vals["Q1100INCL"] = f"{d['inclusive']['ms_per_step']:.2f}"
d = bench("q1_10m")
if d:
    vals["Q110MP"] = f"{100 * d['inclusive']['vs_resident']:.1f}".lstrip("+")
d = bench("c3_int8")
if d:
    vals["C3INT8"] = sp(d["value"])
d = bench("c2_int32")
if d:
    vals["C2INT32"] = sp(d["value"])
d = bench("c4_int32")
if d:
    vals["C4INT32"] = sp(d["value"])
    vals["I32FRAC"] = f"{d['roofline']['valu']['frac']:.2f}"
d = bench("hi")
if d:
    vals["HIINT32"] = f"{d['rerun_ms_per_step']['int32']:.2f}"
t = text("pytest_gpu_tail.txt")
if t:
    m = re.search(r"(\d+) passed, (\d+) skipped", t)
    if m:
        vals["GPUTESTS"] = f"{m.group(1)} passed / {m.group(2)} skipped"
# ---- CLI
t = text("cli_q1_1m_phases.txt")
if t:
    first = re.findall(r"timed region\), total\s+([\d.]+) ms\s*$", t, re.M)
    later = re.findall(r"timed region\), total\s+([\d.]+) ms\s+\(a later pass", t)
    speed = re.findall(r"Search speed:\s+([\d.]+) GCUPS", t)
    if first and speed:
        vals.update(CLIQ1=f"{float(first[0]):.1f}", CLIQ1G=sp(float(speed[0])))
    if later:
        vals["CLIQ1W"] = f"{min(map(float, later)):.1f}–{max(map(float, later)):.1f}"
t = text("cli_1m_phases.txt")
if t:
    first = re.findall(r"timed region\), total\s+([\d.]+) ms", t)
    speed = re.findall(r"Search speed:\s+([\d.]+) GCUPS", t)
    if first and speed:
        vals.update(CLIC4=f"{float(first[0]):.1f}", CLIC4G=sp(float(speed[0])))
t = text("cli_hybrid.txt")
if t:
    part = t.split("== 100 000")[0]
    m0 = re.findall(r"^-m 0: Search speed: ([\d.]+)", part, re.M)
    m1 = [float(x) for x in re.findall(r"^-m 1 -c \d+:.*Search speed: ([\d.]+)", part, re.M)]
    if m0 and m1:
        vals.update(M0C4=sp(float(m0[0])), HYBC4=f"{sp(min(m1))}–{sp(max(m1))}")

p = os.path.join(root, "DESIGN.md")
s = open(template).read()
left = set()
for name in set(re.findall(r"@@([A-Z0-9]+)@@", s)):
    if name in vals:
        s = s.replace(f"@@{name}@@", vals[name])
    else:
        left.add(name)
open(p, "w").write(s)
print("filled:", ", ".join(sorted(vals)))
print("left:", ", ".join(sorted(left)) or "none")
