#!/usr/bin/env python3
"""tools/inclusive_probe.py [NSEQ [QUERY LENGTHS, comma separated]] -- where the time of the upload-inclusive pass goes (default: C2 queries;
chunks of 128 MiB): wall time of every C-ABI call of the pipelined loop bench.py's pcie_inclusive() runs, against the resident pass.
With OSWALD_HIP_DEBUG_SLOW=1 the library adds how the searches lie on the device's time line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oswald_amd import capi, multigpu, submat, synth
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
qlens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else synth.default_query_lengths(); queries = synth.make_queries(qlens)
plan = synth.DatabasePlan(nseq, queries, synth.SEED_DB, 12)
shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "deal")
m = np.array(qlens, dtype=np.uint16); a = np.concatenate(queries); nq = len(qlens)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
ctx = capi.Context(1, [0]); ctx.set_profiling(True); ctx.set_scoring(submat.load("blosum62"), 10, 2, 16); ctx.set_queries(a, m, ad)
chunks = [shard.chunk(k) for k in range(len(shard.mine))]
res = [ctx.chunk_upload(c["b"], c["n"], c["disp"], 16) for c in chunks]
def resident(n):
    for rep in range(n):
        ctx.wait(); t = time.perf_counter()
        for h in res: ctx.chunk_search(h, None)
        ctx.wait(); ms, nl, _ = ctx.kernel_stats(reset=True); print(f"resident pass {1e3*(time.perf_counter()-t):.2f} ms, device time of the {nl} searches {ms:.2f} ms")
resident(3)
if os.environ.get("OSWALD_PROBE_REGISTERED"):   # the residues page-locked in place (oswald_hip_host_register), as the CLI has them, instead of in memory from the library
    class _R:
        def __init__(self, a): self.a = a; self.r = capi.Registered(a)
    class _P:
        def __init__(self, a): self.a = a
    bufs = [[_R(c["b"]), _P(c["n"]), _P(c["disp"]), capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)] for c in chunks]
else:
    bufs = [[capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] + [capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)] for c in chunks]
for rep in range(5):
    if rep == 3: resident(3)
    ctx.wait(); T0 = time.perf_counter(); log = []
    def call(name, f):
        t = time.perf_counter(); r = f(); log.append((name, 1e3 * (t - T0), 1e3 * (time.perf_counter() - t))); return r
    hs = [call("upload0", lambda: ctx.chunk_upload(bufs[0][0].a, bufs[0][1].a, bufs[0][2].a, 16, wait=False))]
    for k in range(len(bufs)):
        call(f"search{k}", lambda: ctx.chunk_search(hs[k], bufs[k][3].a if rep < 2 or rep == 4 else None))
        for j in ((1, 2) if k == 0 else (k + 2,)):
            if j < len(bufs):
                hs.append(call(f"upload{j}", lambda: ctx.chunk_upload(bufs[j][0].a, bufs[j][1].a, bufs[j][2].a, 16, wait=False)))
        call(f"release{k}", lambda: ctx.chunk_release(hs[k]))
    call("wait", lambda: ctx.wait())
    ms, nl, _ = ctx.kernel_stats(reset=True)
    print(f"   device time of the {nl} searches (HIP events around each search's launches): {ms:.2f} ms")
    print(f"inclusive pass {rep} ({'with' if rep < 2 or rep == 4 else 'without'} score tables): {1e3*(time.perf_counter()-T0):.2f} ms")
    for name, at, dur in log: print(f"   {name:10s} at {at:8.2f} ms took {dur:8.2f} ms")
