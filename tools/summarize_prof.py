#!/usr/bin/env python3
"""tools/summarize_prof.py -- condense rocprofv3 CSV output of tools/profile_gpu.sh
into the small files committed under profiles/ (kernel stats, PMC per-launch
averages, HBM traffic per launch of the DP kernel)."""
import csv, glob, hashlib, json, os, sys

out, wl, nseq = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("oswald_amd/csrc/sw_kernels.hip", "oswald_amd/csrc/sw_kernels.h", "oswald_amd/csrc/q8_cell.h", "oswald_amd/csrc/osw_planner.inc")  # = bench.py


def source_digest():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

DP = ("osw_sw_s16qt", "osw_sw_s16q", "osw_sw_s16", "osw_sw_pk16qt", "osw_sw_pk16q", "osw_sw_pk16", "osw_sw_q8")  # first-pass DP kernels (int16 cells; 8-bit cells + their int16 re-run)
RERUN32 = ("osw_sw_i32r", "osw_sw_i32")   # the int32 re-run of a search (round 4: its own kernel, workgroups of twelve waves); cell_bits 32 runs osw_sw_i32
KERNELS = DP + RERUN32 + ("osw_topr", "osw_retile", "osw_block_extent", "osw_build_profile")


def find(sub, pattern):
    r = sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True), key=os.path.getmtime)
    if len(r) > 1:
        print(f"summarize_prof: {len(r)} files match {sub}/{pattern}; taking the newest", file=sys.stderr)
    return r[-1] if r else None


def bench_line(sub):
    """the JSON line bench.py printed in that pass (its stdout is in <sub>.log)"""
    try:
        for line in open(os.path.join(out, sub + ".log")):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except (OSError, ValueError):
        pass
    return None


summary = {"workload": wl, "nseq": int(nseq)}
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows]
f = find("stats", "*kernel_trace.csv")
if f:
    # the library launches every kernel once on empty queues at bring-up (one workgroup): not part of a search
    def one_workgroup(r):
        return int(r.get("Grid_Size_X", "0") or 0) <= int(r.get("Workgroup_Size_X", "256") or 256)
    allrows = [r for r in csv.DictReader(open(f)) if not (r.get("Kernel_Name", "").startswith("osw_sw_") and one_workgroup(r))]
    def kname(r):
        return r.get("Kernel_Name", "").split("(")[0].strip()
    for kn in DP:
        rows = [r for r in allrows if kname(r) == kn]
        if rows:
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            summary[kn + "_trace"] = {"launches": len(d), "avg_ms": sum(d) / len(d) / 1e6, "min_ms": min(d) / 1e6, "max_ms": max(d) / 1e6,
                                      "vgpr": rows[0].get("VGPR_Count"), "sgpr": rows[0].get("SGPR_Count"), "lds": rows[0].get("LDS_Block_Size")}
    # one search step = the pair launch and the single-query launch side by side (two streams) + the int32 re-run:
    # span from the first start to the last end of the k-th dispatches
    dp = {kn: sorted([r for r in allrows if kname(r) == kn], key=lambda r: int(r["Start_Timestamp"])) for kn in DP + RERUN32}
    nstep = max(len(dp[kn]) for kn in RERUN32)   # every chunk search ends with one launch of the re-run kernel
    # the launches of the TIMED steps only: bench.py runs its warm-up steps first and its upload-inclusive passes last
    # (searches into fresh slots, uploads beside them), and times neither with its HIP events
    b0 = bench_line("stats")
    k0, k1 = 0, nstep
    if b0:
        per_step = int(round(b0["roofline"].get("launches_per_step") or 1))
        k0, k1 = per_step * int(b0["warmup"]), min(nstep, per_step * (int(b0["warmup"]) + int(b0["steps"])))
    spans = []
    for k in range(k0, k1):
        rs = [dp[kn][k] for kn in dp if len(dp[kn]) == nstep]
        spans.append(max(int(r["End_Timestamp"]) for r in rs) - min(int(r["Start_Timestamp"]) for r in rs))
    for kn in DP:   # ... and how long a launch of the RESIDENT steps takes (the figure the per-dispatch counters belong to)
        if kn + "_trace" in summary and len(dp[kn]) == nstep and k1 > k0:
            dd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in dp[kn][k0:k1]]
            summary[kn + "_trace"]["resident_launches"] = len(dd)
            summary[kn + "_trace"]["resident_avg_ms"] = sum(dd) / len(dd) / 1e6
    if spans:
        summary["dp_step_span"] = {"steps": len(spans), "dispatch_groups": [k0, k1], "avg_ms": sum(spans) / len(spans) / 1e6, "min_ms": min(spans) / 1e6, "max_ms": max(spans) / 1e6,
                                   "what": "pair kernel + single-query kernel (concurrent) + osw_sw_i32 of one search; compare with bench.py roofline.kernel_ms"}


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    res = {}
    if not f:
        return res
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = next((x for x in KERNELS if k.split("(")[0].strip() == x), None)
        if not name:
            continue
        if name.startswith("osw_sw_") and int(r.get("Grid_Size", "0") or 0) <= int(r.get("Workgroup_Size", "256") or 256):
            continue  # bring-up launch on empty queues (one workgroup)
        c = r["Counter_Name"]
        v = float(r["Counter_Value"])
        e = res.setdefault(name, {}).setdefault(c, {})
        did = int(r.get("Dispatch_Id") or 0)
        e[did] = e.get(did, 0.0) + v
    # The search kernels' dispatches of the RESIDENT steps only (round 6): bench.py runs its warm-up and timed steps first and its
    # upload-inclusive passes -- other launches: a chunk at a time, the first one cut in two -- behind them; until round 5 a "per dispatch"
    # figure averaged over both kinds (and understated what a resident launch moves: the inclusive passes' launches are smaller and more).
    b = bench_line(sub)
    keep = None
    if b:
        keep = int(round((b["roofline"].get("launches_per_step") or 1) * (int(b["warmup"]) + int(b["steps"]))))
    out = {}
    for k, d in res.items():
        out[k] = {}
        for c, byid in d.items():
            ids = sorted(byid)
            if keep and k in DP + RERUN32:
                ids = ids[:keep]
            tot = sum(byid[i] for i in ids)
            out[k][c] = {"sum": tot, "dispatches": len(ids), "per_dispatch": tot / max(1, len(ids))}
    return out


def calib(sub, counter, kernel):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if r.get("Kernel_Name", "").startswith(kernel) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
    return tot or None


# cross-check: the trace's span of one chunk search against the kernel time bench.py measured with
# HIP events IN THE SAME RUN (the JSON line in stats.log); more than 3 % apart = the summary mixes runs, or a kernel is
# missing from the DP list above
b = bench_line("stats")
if b and "dp_step_span" in summary:
    kms = b["roofline"]["kernel_ms"]
    rel = summary["dp_step_span"]["avg_ms"] / kms - 1.0 if kms else None
    summary["consistency"] = {"bench_kernel_ms": kms, "bench_launches_per_step": b["roofline"].get("launches_per_step"), "bench_gcups": b["value"],
                              "trace_step_span_ms": summary["dp_step_span"]["avg_ms"], "relative_difference": rel,
                              "ok": rel is not None and abs(rel) <= 0.03}
summary["pmc_fetch"] = pmc("fetch")
summary["pmc_write"] = pmc("write")
summary["pmc_sq"] = pmc("sq")
summary["pmc_sq2"] = pmc("sq2")
summary["pmc_clk"] = pmc("clk")
# effective core clock under the DP kernels' load: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the duration of THE SAME
# dispatch in the counter pass (its Start / End timestamps; MI355X_MICROARCH.md "DVFS give-back").  (Until the second session of
# round 4 the counter was divided by the kernel's duration in the TRACE pass: a dispatch runs up to 25 % longer under counter
# collection, so that quotient came out anywhere between 2.2 and 2.76 "GHz" for osw_sw_q8 from one box to the next.)
def clk_per_dispatch():
    f = find("clk", "*counter_collection.csv")
    res = {}
    if not f:
        return res
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "").split("(")[0].strip()
        if name not in DP or r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        if int(r.get("Grid_Size", "0") or 0) <= int(r.get("Workgroup_Size", "256") or 256):
            continue  # bring-up launch on empty queues
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        res.setdefault(name, []).append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"]), dur))
    # the dispatches of the resident steps only (see pmc()), and of those the ones of 1 ms and more
    b = bench_line("clk")
    keep = int(round((b["roofline"].get("launches_per_step") or 1) * (int(b["warmup"]) + int(b["steps"])))) if b else None
    out = {}
    for name, lst in res.items():
        lst.sort()
        if keep:
            lst = lst[:keep]
        sel = [(v, dur) for _, v, dur in lst if dur >= 1000000]
        if sel:
            out[name] = sel
    return out
clk = {}
for kn, lst in clk_per_dispatch().items():
    tr = summary.get(kn + "_trace")
    if lst and tr:
        g = [v / 8.0 / d for v, d in lst]
        clk[kn] = {"ghz": sum(g) / len(g), "ghz_min": min(g), "ghz_max": max(g), "dispatches": len(g),
                   "counter_pass_avg_ms": sum(d for _, d in lst) / len(lst) / 1e6, "trace_avg_ms": tr["avg_ms"]}
if clk:
    summary["effective_clock"] = clk
    dom = max(clk, key=lambda k: clk[k]["trace_avg_ms"])
    sq = summary.get("pmc_sq", {}).get(dom, {})
    nv = sq.get("SQ_INSTS_VALU", {}).get("per_dispatch")
    if nv:
        # VALU issue rate of the dominant kernel in cycles of the clock it actually ran at, per SIMD (1024 SIMDs); 2.0 = the
        # SIMD-32 raw issue rate of a wave64 instruction (MI355X_MICROARCH.md)
        ms = (summary.get(dom + "_trace") or {}).get("resident_avg_ms") or clk[dom]["trace_avg_ms"]  # (a launch of the resident steps: what `nv` is counted over)
        cyc = clk[dom]["ghz"] * 1e9 * ms * 1e-3 * 1024.0 / nv
        summary["valu_issue"] = {"kernel": dom, "valu_wave_instructions": nv, "launch_ms": ms, "cycles_per_instruction_per_simd_at_effective_clock": cyc,
                                 "cycles_per_instruction_per_simd_at_2p4GHz": 2.4e9 * ms * 1e-3 * 1024.0 / nv,
                                 "raw_issue_fraction": 2.0 / cyc}
try:
    fs = sum(summary["pmc_fetch"].get(k, {}).get("FETCH_SIZE", {}).get("per_dispatch", 0.0) for k in DP + RERUN32)
    ws = sum(summary["pmc_write"].get(k, {}).get("WRITE_SIZE", {}).get("per_dispatch", 0.0) for k in DP + RERUN32)
    if fs == 0 and ws == 0:
        raise KeyError("no DP kernel counters")
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 64 B per
    # 128-B request for wide coalesced reads, other widths must be calibrated on a known byte count in the kernel's own
    # access pattern.  The DP kernel moves 8 B per lane; `ubench calib` reads and writes exactly 1 GiB that way.
    GIB = float(1 << 30)
    # (two dword accesses per 8-B entry, as the kernels' column loop issues them; the 8-B variant is kept for comparison)
    cf = calib("calib_fetch", "FETCH_SIZE", "stream_read4x2") or calib("calib_fetch", "FETCH_SIZE", "stream_read8")
    cw = calib("calib_write", "WRITE_SIZE", "stream_write4x2") or calib("calib_write", "WRITE_SIZE", "stream_write8")
    summary["calib_8B_access"] = {"fetch_kib_for_1GiB": calib("calib_fetch", "FETCH_SIZE", "stream_read8"), "write_kib_for_1GiB": calib("calib_write", "WRITE_SIZE", "stream_write8")}
    kf = GIB / (cf * 1024) if cf else 2.0
    kw = GIB / (cw * 1024) if cw else 1.0
    summary["traffic"] = {"fetch_kib_raw": fs, "write_kib_raw": ws, "calib_fetch_kib_for_1GiB": cf, "calib_write_kib_for_1GiB": cw,
                          "fetch_factor": kf, "write_factor": kw,
                          "hbm_bytes_per_launch": int((kf * fs + kw * ws) * 1024),
                          "correction": "bytes = KiB * 1024 * factor; factor = 1 GiB / counter value of a 1-GiB stream of 8-B entries accessed as 2 dwords per lane (the DP kernels' spill access)"}
    # what bench.py reads for roofline.traffic: tied to the kernel sources it was measured on
    # ... and to the MODE it was measured in (round 6: a file is keyed by workload, database size, cell arithmetic and whether the pairs' tails run)
    tails = os.environ.get("OSWALD_HIP_PAIR_TAILS", "1")
    t = dict(summary["traffic"], workload=wl, nseq=int(nseq), source_digest=source_digest(), dtype=(b or {}).get("dtype"), pair_tails=tails,
             algorithmic_bytes_per_launch=((b or {}).get("roofline") or {}).get("algorithmic_bytes_per_launch"), launches_per_step=((b or {}).get("roofline") or {}).get("launches_per_step"),
             source=f"tools/profile_gpu.sh {wl} {nseq} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), bench.py's default cells for the workload, OSWALD_HIP_PAIR_TAILS={tails}")
    with open(os.path.join(out, f"traffic_{wl}_{nseq}" + ("" if tails == "1" else f"_tails{tails}") + ".json"), "w") as f:
        json.dump(t, f, indent=1)
except KeyError:
    pass
print(json.dumps(summary, indent=1))
