#!/usr/bin/env python3
"""tools/plan_probe2.py -- why is the first search of an upload-inclusive pass faster than the same search of a resident chunk?
Score tables of the pipelined flow against those of resident searches, and the first search's device time in variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
qlens = synth.default_query_lengths(); queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_profiling(True); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
bufs = [[capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] + [capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)] for c in chunks]
truth = []
for c in chunks:
    h = ctx.chunk_upload(c["b"], c["n"], c["disp"], 16)
    out = np.zeros((nq, len(c["n"]) * 16), np.int32)
    ctx.chunk_search(h, out); ctx.wait(); ctx.chunk_release(h); truth.append(out)
def t_first(order, tables=True):
    """order: list of ('u', k) / ('s', k) steps; returns the device times of the searches and whether the tables are right"""
    ctx.wait(); ctx.kernel_stats(reset=True)
    for b in bufs: b[3].a[:] = -7
    hs = {}
    times = []
    t0 = time.perf_counter(); log = []
    for op, k in order:
        t1 = time.perf_counter()
        if op == "u": hs[k] = ctx.chunk_upload(bufs[k][0].a, bufs[k][1].a, bufs[k][2].a, 16, wait=False)
        elif op == "s":
            ctx.chunk_search(hs[k], bufs[k][3].a if tables else None); log.append((op + str(k), round(1e3 * (t1 - t0), 1), round(1e3 * (time.perf_counter() - t1), 1))); t1 = time.perf_counter(); op = "r"; ctx.chunk_release(hs[k])
        elif op == "w":
            ctx.wait(); times.append(ctx.kernel_stats(reset=True)[0])
        log.append((op + str(k), round(1e3 * (t1 - t0), 1), round(1e3 * (time.perf_counter() - t1), 1)))
    t1 = time.perf_counter(); ctx.wait(); log.append(("wait", round(1e3 * (t1 - t0), 1), round(1e3 * (time.perf_counter() - t1), 1)))
    times.append(ctx.kernel_stats(reset=True)[0]); times.append(("wall", round(1e3 * (time.perf_counter() - t0), 2))); times.append(log)
    ok = [bool(np.array_equal(bufs[k][3].a, truth[k])) for op, k in order if op == "s"] if tables else None
    return [round(t, 2) if isinstance(t, float) else t for t in times], ok
def run_topr(order):
    ctx.wait(); ctx.kernel_stats(reset=True); ctx.topr_begin(10)
    hs = {}
    t0 = time.perf_counter()
    for op, k in order:
        if op == "u": hs[k] = ctx.chunk_upload(bufs[k][0].a, bufs[k][1].a, bufs[k][2].a, 16, wait=False); ctx.chunk_set_index(hs[k], 0, chunks[k]["nseq"], chunks[k]["gpos"])
        elif op == "s": ctx.chunk_search(hs[k], None); ctx.chunk_release(hs[k])
    sc, ix = ctx.topr(10)
    return round(1e3 * (time.perf_counter() - t0), 2), sc, ix
# ground truth of the top lists: resident chunks, synchronous uploads
res = []
ctx.topr_begin(10)
for c in chunks:
    h = ctx.chunk_upload(c["b"], c["n"], c["disp"], 16); ctx.chunk_set_index(h, 0, c["nseq"], c["gpos"]); ctx.chunk_search(h, None); res.append(h)
sc0, ix0 = ctx.topr(10)
for h in res: ctx.chunk_release(h)
bench_order = [("u", 0), ("s", 0), ("u", 1), ("u", 2), ("s", 1), ("s", 2)]
probe_order = [("u", 0), ("u", 1), ("s", 0), ("u", 2), ("s", 1), ("s", 2)]
for rep in range(3):
    print("tables, bench order ", t_first(bench_order))
    print("tables, probe order ", t_first(probe_order))
    for name, o in (("bench", bench_order), ("probe", probe_order)):
        ms, sc, ix = run_topr(o)
        print(f"top lists, {name} order: {ms} ms, equal to the resident run's: {bool(np.array_equal(sc, sc0) and np.array_equal(ix, ix0))}")
