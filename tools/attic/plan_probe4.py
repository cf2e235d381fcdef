#!/usr/bin/env python3
"""tools/plan_probe4.py WORKLOAD NSEQ -- one-chunk upload-inclusive passes, call by call (wall ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
wl, nseq = sys.argv[1], int(sys.argv[2])
qlens = {"q1": [375], "c5": [5000]}.get(wl) or synth.default_query_lengths()
queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
c = shard.chunk(0)
bufs = [capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] + [capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)]
for rep in range(6):
    ctx.wait(); t0 = time.perf_counter(); log = []
    def call(name, f):
        t = time.perf_counter(); r = f(); log.append(f"{name}:{1e3*(time.perf_counter()-t):.2f}"); return r
    h = call("upload", lambda: ctx.chunk_upload(bufs[0].a, bufs[1].a, bufs[2].a, 16, wait=False))
    call("search", lambda: ctx.chunk_search(h, bufs[3].a))
    call("release", lambda: ctx.chunk_release(h))
    call("wait", lambda: ctx.wait())
    print(f"pass {rep}: {1e3*(time.perf_counter()-t0):.2f} ms  " + " ".join(log), flush=True)
