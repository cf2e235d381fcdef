"""N > 1 path on CPUs: two processes over gloo, each owning the chunks the
reference's round-robin would give its device; per-rank top-r lists are
all-gathered and merged with the reference's tie rule.  The per-rank scores
come from the CPU oracle here (no GPU), so this covers sharding, global index
bookkeeping, the collective and the merge."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oswald_amd import dblayout, multigpu, submat, synth

from helpers import pack_queries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, r, ret):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs = synth.make_queries([60, 35, 90], seed=5)
    L, R, O = synth.make_database(900, qs, seed=21, homologs_per_query=5)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    n_all = dblayout.group_lengths(sl, 16)
    a, m, ad = pack_queries(qs)
    sm = submat.load("blosum62")
    mine = multigpu.rank_chunks(n_all, 16, 134217728, world, rank)
    best_s = np.full((3, r), -1, np.int32)
    best_i = np.full((3, r), -1, np.int64)
    for g0, g1 in mine:
        b, n, disp = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
        sc = pyoracle.search_chunk_scalar(a, m, ad, b, n, disp.astype(np.uint32), 16, sm, 10, 2, threads=2)
        nvalid = min(900, g1 * 16) - g0 * 16
        for q in range(3):
            s_, i_ = dblayout.topr_reference_order(sc[q, :nvalid], r)
            s2, i2 = dblayout.merge_topr([(best_s[q], best_i[q]), (s_, i_.astype(np.int64) + g0 * 16)], r)
            best_s[q, :len(s2)], best_i[q, :len(i2)] = s2, i2
    out_s, out_i = multigpu.gather_topr(best_s, best_i, r, dist)
    if rank == 0:
        ret["scores"], ret["index"], ret["chunks"] = out_s, out_i, mine
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_merge(oracle):
    world, r = 2, 12
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), r, ret), nprocs=world, join=True)
    qs = synth.make_queries([60, 35, 90], seed=5)
    L, R, O = synth.make_database(900, qs, seed=21, homologs_per_query=5)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    a, m, ad = pack_queries(qs)
    whole = oracle.search_chunk_scalar(a, m, ad, b, n, disp.astype(np.uint32), 16, submat.load("blosum62"), 10, 2)
    for q in range(3):
        ws, wi = dblayout.topr_reference_order(whole[q, :900], r)
        np.testing.assert_array_equal(ret["scores"][q], ws)
        np.testing.assert_array_equal(ret["index"][q], wi)
    assert len(ret["chunks"]) >= 1


def test_rank_chunks_cover_database_once():
    L, R, O = synth.make_database(5000, seed=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    n = dblayout.group_lengths(sl, 16)
    for world in (1, 2, 4, 8):
        seen = []
        sizes = []
        for rank in range(world):
            ch = multigpu.rank_chunks(n, 16, 134217728, world, rank)
            seen += ch
            sizes.append(sum(int(n[g0:g1].sum()) * 16 for g0, g1 in ch))
        seen.sort()
        assert seen[0][0] == 0 and seen[-1][1] == len(n)
        assert all(seen[i][1] == seen[i + 1][0] for i in range(len(seen) - 1))
        if world > 1:
            assert max(sizes) < 1.35 * (sum(sizes) / world)  # equal padded residues, last shard smaller
