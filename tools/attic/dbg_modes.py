"""Debug aid: full-size C2 search with int16 and with int32 cells, report where they differ (run on the GPU box;
the int32 pass takes a few seconds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oswald_amd import capi, submat, synth, dblayout
from helpers import pack_queries

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
qs = synth.make_queries(synth.default_query_lengths())
L, R, O = synth.make_database(nseq, qs)
order, sl, sr, so = dblayout.sort_by_length(L, R, O)
b, n, disp = dblayout.interleave(sl, sr, so, 16)
a, m, ad = pack_queries(qs)
sm = submat.load("blosum62")
ctx = capi.Context(1)
res = {}
for bits in (16, 32):
    ctx.set_scoring(sm, 10, 2, bits)
    ctx.set_queries(a, m, ad)
    out = np.zeros((len(qs), len(n) * 16), np.int32)
    ctx.search_chunk_async(b, n, disp.astype(np.uint32), out, 16)
    ctx.wait()
    res[bits] = out
    print(bits, "rerun items", ctx.kernel_stats()[2], "max", out.max())
bad = np.argwhere(res[16] != res[32])
print("mismatches", len(bad))
for q, s in bad[:20]:
    print("  q", q, "m", m[q], "seq", s, "len", sl[s] if s < nseq else -1, "i16", res[16][q, s], "i32", res[32][q, s])
if len(bad):
    print("queries", sorted(set(bad[:, 0].tolist())))
    print("blocks", sorted(set((bad[:, 1] // 128).tolist()))[:40])
