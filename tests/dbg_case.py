"""Debug aid: small GPU-vs-oracle comparisons at forced geometries (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oswald_amd import capi, submat, synth
from oracle import pyoracle
from helpers import layout, pack_queries, random_db

def run(qlens, nseq, maxlen, env, bits=0):
    for k in ("OSWALD_HIP_FORCE_LG", "OSWALD_HIP_FORCE_WG", "OSWALD_HIP_PAIRS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    qs = synth.make_queries(qlens, seed=3)
    L, R, O = random_db(nseq, seed=5, max_len=maxlen, queries=qs, homologs=1)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    ctx = capi.Context()
    ctx.set_scoring(sm, 10, 2, bits)
    ctx.set_queries(a, m, ad)
    out = np.full((len(qs), len(n) * 16), -7, dtype=np.int32)
    ctx.search_chunk_async(b, n, disp, out, 16)
    ctx.wait()
    ctx.close() if hasattr(ctx, "close") else None
    want = pyoracle.search_chunk_scalar(a, m, ad, b, n, disp, 16, sm, 10, 2)
    bad = np.argwhere(out != want)
    print(qlens, nseq, maxlen, env, bits, "mismatches", len(bad), "of", out.size)
    if len(bad):
        for q, s in bad[:6]:
            print("   q", q, "seq", s, "got", out[q, s], "want", want[q, s])

if __name__ == "__main__":
    run([20], 8, 30, {"OSWALD_HIP_FORCE_LG": "0", "OSWALD_HIP_PAIRS": "0"})
    run([20], 8, 30, {"OSWALD_HIP_FORCE_LG": "0", "OSWALD_HIP_PAIRS": "0"}, bits=32)
    run([20], 200, 60, {"OSWALD_HIP_FORCE_LG": "0", "OSWALD_HIP_PAIRS": "0"})
    run([20], 200, 60, {"OSWALD_HIP_FORCE_LG": "1", "OSWALD_HIP_PAIRS": "0"})
    run([100], 200, 60, {"OSWALD_HIP_FORCE_LG": "0", "OSWALD_HIP_PAIRS": "0"})
    run([100], 200, 60, {"OSWALD_HIP_FORCE_LG": "2", "OSWALD_HIP_PAIRS": "0"})
    run([100, 100], 200, 60, {"OSWALD_HIP_FORCE_LG": "0", "OSWALD_HIP_PAIRS": "2"})
    run([300], 200, 60, {"OSWALD_HIP_FORCE_LG": "2", "OSWALD_HIP_FORCE_WG": "1", "OSWALD_HIP_PAIRS": "0"})
