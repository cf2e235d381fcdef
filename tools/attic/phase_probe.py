#!/usr/bin/env python3
"""tools/phase_probe.py WORKLOAD NSEQ -- host-side phase times (OSWALD_HIP_DEBUG_PHASES) of one upload + search of every
chunk of a synthetic database, from pageable numpy arrays."""
import os, sys, time
os.environ["OSWALD_HIP_DEBUG_PHASES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
wl, nseq = sys.argv[1], int(sys.argv[2])
qlens = {"q1": [375], "c5": [5000]}.get(wl) or synth.default_query_lengths()
queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
for rep in range(2):
    print("--- pass", rep, file=sys.stderr)
    for c in chunks:
        t = time.perf_counter()
        h = ctx.chunk_upload(c["b"], c["n"], c["disp"], 16)
        t1 = time.perf_counter()
        ctx.chunk_search(h, None); ctx.wait()
        t2 = time.perf_counter()
        ctx.chunk_release(h)
        print(f"chunk: upload {1e3*(t1-t):.2f} ms, search {1e3*(t2-t1):.2f} ms, {len(c['n'])} groups, {c['b'].size} bytes", file=sys.stderr)
