"""Sanity: RCCL (torch.distributed 'nccl', one rank) and liboswald_hip.so in one process on one GPU --
the collectives bench.py uses for N > 1 (all_reduce MAX/SUM, all_gather of the top-r lists, barrier)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, torch.distributed as dist
from oswald_amd import capi, multigpu, submat, synth, dblayout

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
qs = synth.make_queries([120, 300], seed=3)
L, R, O = synth.make_database(3000, qs)
order, sl, sr, so = dblayout.sort_by_length(L, R, O)
b, n, disp = dblayout.interleave(sl, sr, so, 16)
m = np.array([len(q) for q in qs], np.uint16); a = np.concatenate(qs); ad = np.array([0, len(qs[0])], np.uint32)
ctx = capi.Context(1, [0]); ctx.set_scoring(submat.load("blosum62"), 10, 2); ctx.set_queries(a, m, ad)
h = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16); ctx.chunk_search(h, None)
sc, ix = ctx.chunk_topr(h, 3000, 5)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
top = multigpu.gather_topr(sc, ix.astype(np.int64), 5, dist, dev)
print("nccl smoke ok:", top[0][:, 0].tolist(), float(t.item()))
ctx.close(); dist.destroy_process_group()
