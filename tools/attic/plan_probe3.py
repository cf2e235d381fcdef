#!/usr/bin/env python3
"""tools/plan_probe3.py -- which sequence of calls makes a search's plan wait for the running search (OSWALD_HIP_DEBUG_SLOW)?"""
import os, sys, time
os.environ["OSWALD_HIP_DEBUG_SLOW"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
nseq = 1000000
qlens = synth.default_query_lengths(); queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_profiling(True); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
bufs = [[capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] for c in chunks]
def run(name, order):
    for rep in range(2):
        ctx.wait(); hs = {}; log = []; t0 = time.perf_counter()
        for step in order.split():
            op, k = step[0], int(step[1]); t1 = time.perf_counter()
            if op == "u": hs[k] = ctx.chunk_upload(bufs[k][0].a, bufs[k][1].a, bufs[k][2].a, 16, wait=False)
            elif op == "s": ctx.chunk_search(hs[k], None)
            elif op == "r": ctx.chunk_release(hs[k])
            log.append(f"{step}:{1e3*(time.perf_counter()-t1):.1f}")
        ctx.wait(); wall = 1e3 * (time.perf_counter() - t0); ctx.kernel_stats(reset=True)
        for k in list(hs):
            try: ctx.chunk_release(hs[k])
            except Exception: pass
        print(f"{name:28s} pass {rep}: {wall:7.1f} ms  " + " ".join(log), flush=True)
        sys.stderr.flush()
run("A: no look-ahead", "u0 s0 r0 u1 s1 r1")
run("B: bench order", "u0 s0 r0 u1 u2 s1 r1 s2 r2")
run("C: no release", "u0 s0 u1 s1")
run("D: probe order", "u0 u1 s0 r0 u2 s1 r1 s2 r2")
run("E: slot re-use only", "u0 s0 r0 u2 s2 r2")
run("F: u1 behind s0, no u2", "u0 s0 r0 u1 s1 r1 u2 s2 r2")
