#!/usr/bin/env python3
"""tools/plan_probe.py [NSEQ] -- device time of every chunk search (C2 queries, 128-MiB chunks) when the chunk is
(a) resident, planned on the live extents; (b) uploaded again, waited for, planned on the live extents; (c) uploaded
asynchronously and searched at once (planned on the group-length extents, queued behind the upload)."""
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
bufs = [[capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] for c in chunks]
def one(h):
    ctx.wait(); ctx.kernel_stats(reset=True)
    ctx.chunk_search(h, None); ctx.wait()
    return ctx.kernel_stats(reset=True)[0]
res = [ctx.chunk_upload(c["b"], c["n"], c["disp"], 16) for c in chunks]
for rep in range(3):
    print("resident, live extents:      ", " ".join(f"{one(h):8.2f}" for h in res), flush=True)
for rep in range(3):
    out = []
    for k in range(len(bufs)):
        h = ctx.chunk_upload(bufs[k][0].a, bufs[k][1].a, bufs[k][2].a, 16, wait=True)
        out.append(one(h)); ctx.chunk_release(h)
    print("uploaded + waited, live:     ", " ".join(f"{x:8.2f}" for x in out), flush=True)
for rep in range(3):
    out = []
    for k in range(len(bufs)):
        ctx.wait(); ctx.kernel_stats(reset=True)
        h = ctx.chunk_upload(bufs[k][0].a, bufs[k][1].a, bufs[k][2].a, 16, wait=False)
        ctx.chunk_search(h, None); ctx.chunk_release(h); ctx.wait()
        out.append(ctx.kernel_stats(reset=True)[0])
    print("uploaded async, group-length:", " ".join(f"{x:8.2f}" for x in out), flush=True)
for rep in range(2):
    print("resident again:              ", " ".join(f"{one(h):8.2f}" for h in res), flush=True)
os.environ["X"] = "1"
