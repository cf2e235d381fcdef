#!/usr/bin/env python3
"""tools/tail_overlap_probe.py -- what would overlapping the tail of one chunk search with the start of the next be worth?
Two context devices on ONE GPU have streams, queue counters and spill scratch of their own: chunks searched alternately on
them run as persistent grids side by side, the second one's workgroups taking the slots the first one's leave."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
qlens = {"q1": [375], "c5": [5000]}.get(wl) or synth.default_query_lengths()
queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
ctx = capi.Context(2, [0, 0]); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
for name, devs in (("all on one device", [0] * len(chunks)), ("alternating devices", [k % 2 for k in range(len(chunks))])):
    hs = [ctx.chunk_upload(c["b"], c["n"], c["disp"], 16, dev=d) for c, d in zip(chunks, devs)]
    for rep in range(4):
        ctx.wait(); t0 = time.perf_counter()
        for h, d in zip(hs, devs): ctx.chunk_search(h, None, dev=d)
        ctx.wait()
        print(f"{name}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    for h, d in zip(hs, devs): ctx.chunk_release(h, dev=d)
