#!/usr/bin/env python3
"""tools/kbench.py -- steady-state rate of the DP kernel on a uniform workload
(all sequences the same length, all queries the same length): no load
imbalance, no tail; what the inner loop itself sustains."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oswald_amd import capi, dblayout, submat, synth

ap = argparse.ArgumentParser()
ap.add_argument("--nseq", type=int, default=131072)
ap.add_argument("--len", type=int, default=360)
ap.add_argument("--nq", type=int, default=8)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--bits", type=int, default=16)
a_ = ap.parse_args()

L = np.full(a_.nseq, a_.len, dtype=np.uint16)
O = np.arange(a_.nseq + 1, dtype=np.int64) * a_.len
R = synth.random_residues(1, 0, a_.nseq * a_.len)
b, n, disp = dblayout.interleave(L, R, O, 16, round_to=4)
qs = synth.make_queries([a_.m] * a_.nq)
m = np.array([a_.m] * a_.nq, np.uint16)
ad = (np.arange(a_.nq) * a_.m).astype(np.uint32)
ctx = capi.Context(1)
ctx.set_scoring(submat.load("blosum62"), 10, 2, a_.bits)
ctx.set_queries(np.concatenate(qs), m, ad)
h = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
ctx.chunk_search(h, None); ctx.wait()
ctx.set_profiling(True); ctx.kernel_stats(reset=True)
for _ in range(a_.reps):
    ctx.chunk_search(h, None)
ctx.wait()
ms, nl, _ = ctx.kernel_stats()
cells = float(a_.nq) * a_.m * a_.nseq * a_.len
g = ctx.chunk_geometry(h)
print(f"items={g['work_items']} maxlg={g['max_log2_geometry']}", end=" ")
print(f"uniform nseq={a_.nseq} len={a_.len} nq={a_.nq} m={a_.m} bits={a_.bits}: {ms/nl:.3f} ms/launch, {cells/(ms/nl*1e-3)/1e9:.1f} GCUPS")
