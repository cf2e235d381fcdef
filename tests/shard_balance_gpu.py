"""Diagnostic (GPU box): time every rank's shard of the C4 database on ONE GPU, one after the other, to predict the
load balance of a shard rule (RULE=deal|reference, default both) at N ranks: efficiency = mean(shard time) / max(shard
time); the predicted speed-up is against the same database searched whole on the one GPU (measured first).
    python tests/shard_balance_gpu.py [world ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oswald_amd import capi, multigpu, submat, synth

worlds = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
max_chunk = int(os.environ.get("MAX_CHUNK", "134217728"))
qlens = synth.default_query_lengths()
qs = synth.make_queries(qlens)
m = np.array(qlens, np.uint16)
a = np.concatenate(qs)
ad = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
plan = synth.DatabasePlan(1000000, qs, synth.SEED_DB, 12)
ctx = capi.Context(1)
ctx.set_scoring(submat.load("blosum62"), 10, 2)
ctx.set_queries(a, m, ad)
rules = [os.environ["RULE"]] if os.environ.get("RULE") else ["deal", "reference"]
t_whole = None
for rule, world in [("deal", 1)] + [(r, w) for r in rules for w in worlds]:
    times, res = [], []
    only = os.environ.get("ONLY_RANK")
    for rank in ([int(only)] if only is not None and world > 1 else range(world)):
        sh = multigpu.ShardedDatabase(plan, 16, max_chunk, world, rank, rule)
        chunks = [sh.chunk(k) for k in range(len(sh.mine))]
        hs = [ctx.chunk_upload(c["b"], c["n"], c["disp"], 16) for c in chunks]
        def step():
            for h in hs: ctx.chunk_search(h, None)
            for h, c in zip(hs, chunks): ctx.chunk_topr(h, c["nseq"], 10)
        step()
        t0 = time.perf_counter()
        for _ in range(3): step()
        times.append((time.perf_counter() - t0) / 3)
        res.append(sum(int(c["off"][-1]) for c in chunks))
        for h in hs: ctx.chunk_release(h)
    tot = float(m.astype(np.int64).sum()) * sum(res)
    if world == 1:
        t_whole = times[0]
        print(f"whole database on one GPU ({len(sh.mine)} chunks): {t_whole * 1e3:.1f} ms = {tot / t_whole / 1e9:.0f} GCUPS", flush=True)
        continue
    print(f"rule {rule} max_chunk {max_chunk} chunks/rank {len(sh.mine)} world {world}: shard ms {[round(t * 1e3, 1) for t in times]}  residues {[round(r / 1e6, 1) for r in res]} M  "
          f"-> predicted {tot / max(times) / 1e9:.0f} GCUPS = {t_whole / max(times):.2f} x the whole database on one GPU, balance {np.mean(times) / max(times):.3f}", flush=True)
ctx.close()
