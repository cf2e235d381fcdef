#!/usr/bin/env python3
"""tools/inclusive_steps.py -- which step of a short search's PCIe-inclusive pass is the slow one (LABNOTES, third session of round 4 (6))?
One 375-residue query, 100 000 sequences, page-locked buffers: upload alone, search alone (chunk resident) with and without the score
table's download, the whole pass; the same with pageable buffers.  Times of 6 repetitions each, ms."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oswald_amd import capi, dblayout, submat, synth
qs = synth.make_queries([375])
L, R, O = synth.make_database(100000, qs)
order, sl, sr, so = dblayout.sort_by_length(L, R, O)
b, n, disp = dblayout.interleave(sl, sr, so, 16)
disp = disp.astype(np.uint32)
m = np.array([375], np.uint16); ad = np.zeros(1, np.uint32); a = qs[0]
def ms(f, reps=6):
    out = []
    for _ in range(reps):
        t = time.perf_counter(); f(); out.append(round((time.perf_counter() - t) * 1e3, 2))
    return out
with capi.Context(1) as ctx:
    ctx.set_scoring(submat.load("blosum62"), 10, 2, 0)
    ctx.set_queries(a, m, ad)
    for kind in ("page-locked", "pageable"):
        if kind == "page-locked":
            keep = [capi.pinned_copy(x) for x in (b, n, disp)] + [capi.HostBuffer((1, len(n) * 16), np.int32)]
            hb, hn, hd, out = keep[0].a, keep[1].a, keep[2].a, keep[3].a
        else:
            hb, hn, hd, out = b, n, disp, np.zeros((1, len(n) * 16), np.int32)
        def up():
            h = ctx.chunk_upload(hb, hn, hd, 16, wait=False); ctx.wait(); ctx.chunk_release(h)
        h0 = ctx.chunk_upload(hb, hn, hd, 16)
        def search_table():
            ctx.chunk_search(h0, out); ctx.wait()
        def search_only():
            ctx.chunk_search(h0, None); ctx.wait()
        def whole():
            h = ctx.chunk_upload(hb, hn, hd, 16, wait=False); ctx.chunk_search(h, out); ctx.chunk_release(h); ctx.wait()
        for name, f in (("upload + wait", up), ("search (resident), no table", search_only), ("search (resident) + table", search_table), ("upload + search + table", whole)):
            f()
            print(f"{kind:12s} {name:32s}", ms(f), flush=True)
        ctx.chunk_release(h0)
        if kind == "page-locked":
            for k in keep: k.close()
