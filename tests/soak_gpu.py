"""Soak: many searches with changing query sets against one resident database, every score checked against the
CPU port (oracle), plus repeated identical searches (determinism).  Test infrastructure (it checks against oracle/, so it lives under tests/).  Run on the GPU box: python tests/soak_gpu.py [iters]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oswald_amd import capi, submat, synth, dblayout
from oracle import pyoracle
from helpers import pack_queries

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nseq = 20000
rng = np.random.default_rng(12345)
base_q = synth.make_queries(synth.default_query_lengths())
L, R, O = synth.make_database(nseq, base_q)
order, sl, sr, so = dblayout.sort_by_length(L, R, O)
b, n, disp = dblayout.interleave(sl, sr, so, 16)
disp = disp.astype(np.uint32)
mats = ["blosum62", "pam250", "blosum45", "blosum80"]
ctx = capi.Context(1)
h = None
parts_h = None
t0 = time.time()
bad = 0
cells = 0
for it in range(iters):
    nq = int(rng.integers(1, 9))
    lens = [int(x) for x in rng.integers(1, int(rng.choice([60, 300, 1200])) + 1, nq)]
    qs = [synth.random_residues(1000 * it + k, 0, m) for k, m in enumerate(lens)]
    if it % 3 == 0:  # some real hits: pieces of database sequences
        s = int(rng.integers(nseq // 2, nseq)); seg = sr[so[s]:so[s] + sl[s]]
        qs[0] = np.ascontiguousarray(seg[: max(1, min(len(seg), lens[0]))])
    a, m, ad = pack_queries(qs)
    sm = submat.load(mats[it % len(mats)])
    go, ge = int(rng.integers(0, 20)), int(rng.integers(0, 6))
    bits = [0, 16, 32, 8][(it // 3) % 4] if it % 3 == 2 else 0   # every third search in an explicit cell mode, 8-bit included
    os.environ["OSWALD_HIP_PAIR_TAILS"] = ["2", "1", "0"][it % 3]   # every eligible pair item with its tail / the cost model's choice / padded pairs (read at set_scoring)
    ctx.set_scoring(sm, go, ge, bits)
    ctx.set_queries(a, m, ad)
    if h is None:
        h = ctx.chunk_upload(b, n, disp, 16)
    out = np.zeros((len(qs), len(n) * 16), np.int32)
    ctx.chunk_search(h, out); ctx.wait()
    out2 = np.zeros_like(out)
    if it % 2:
        ctx.chunk_search(h, out2); ctx.wait()
    else:
        # the same database as three resident chunks searched as ONE launch (oswald_hip_search_resident): the same table, columns side by side
        if parts_h is None:
            plan3 = dblayout.chunk_plan(n, 16, int(b.size) // 3 + 1, 1)
            parts_h = []
            for g0, g1 in plan3:
                pb, pn, pd = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
                parts_h.append(ctx.chunk_upload(pb, pn, pd.astype(np.uint32), 16))
            assert len(parts_h) >= 3
        ctx.search_resident(parts_h, out2); ctx.wait()
    want, _ = pyoracle.search_chunk_simd(a, m, ad, b, n, disp, 16, sm, go, ge)
    ok = np.array_equal(out, want) and np.array_equal(out, out2)
    cells += int(m.astype(np.int64).sum()) * int(sl.astype(np.int64).sum())
    if not ok:
        bad += 1
        w = np.argwhere(out != want)
        print("MISMATCH it", it, "lens", lens, "go/ge", go, ge, "bits", bits, "n", len(w), "first", w[:3].tolist(), flush=True)
    if it % 20 == 0:
        print(f"it {it}: {bad} bad, {cells/1e9:.0f} Gcells checked, {time.time()-t0:.0f} s", flush=True)
print("soak done:", iters, "iterations,", bad, "mismatching")
sys.exit(1 if bad else 0)
