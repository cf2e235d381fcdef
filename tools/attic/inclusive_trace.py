#!/usr/bin/env python3
"""tools/inclusive_trace.py [NSEQ] -- two upload-inclusive passes (the loop of bench.py's pcie_inclusive) for a rocprofv3
--kernel-trace --memory-copy-trace run: what the device does between two searches.  Prints the wall-clock (ns, CLOCK_MONOTONIC is
not the profiler's clock: the passes are found in the trace by their kernels) and the host-side phase laps."""
import os, sys, time
os.environ.setdefault("OSWALD_HIP_DEBUG_PHASES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
tables = len(sys.argv) > 2 and sys.argv[2] == "tables"
qlens = synth.default_query_lengths(); queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
bufs = [[capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] + [capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)] for c in chunks]
for rep in range(3):
    ctx.wait(); T0 = time.perf_counter()
    print(f"=== pass {rep}", file=sys.stderr, flush=True)
    hs = [ctx.chunk_upload(bufs[0][0].a, bufs[0][1].a, bufs[0][2].a, 16, wait=False)]
    for k in range(len(bufs)):
        ctx.chunk_search(hs[k], bufs[k][3].a if tables else None)
        ctx.chunk_release(hs[k])
        for j in ((1, 2) if k == 0 else (k + 2,)):
            if j < len(bufs):
                hs.append(ctx.chunk_upload(bufs[j][0].a, bufs[j][1].a, bufs[j][2].a, 16, wait=False))
    ctx.wait()
    print(f"inclusive pass {rep}: {1e3*(time.perf_counter()-T0):.2f} ms", flush=True)
