"""Sharding of the search path over GPUs: one process per GPU, every rank owns
one contiguous shard of the length-sorted database (the reference gives chunk
c of a round to device c mod ndev, host/src/FPGAsearch.c:132-138, and merges
by memcpy, :236-237); the only exchange is the gather of the per-rank top-r
lists, which is what runs over RCCL/xGMI (backend "nccl") or gloo on CPUs.

Plumbing only: torch.distributed carries (score, global index) pairs; the
ordering rule is the reference's (descending score, ties by descending index,
host/src/utils.c:3-86)."""
from __future__ import annotations

import numpy as np

from . import dblayout


def rank_chunks(n_groups_len, W: int, max_chunk_size: int, world: int, rank: int):
    """Group ranges this rank searches: the database is cut by the reference's
    rule (dblayout.chunk_plan, host/src/sequences.c:505-541) and chunk c goes to
    device c mod world (host/src/FPGAsearch.c:132-138)."""
    plan = dblayout.chunk_plan(n_groups_len, W, max_chunk_size, world)
    return [p for c, p in enumerate(plan) if c % world == rank]


class ShardedDatabase:
    """A synthetic database (synth.DatabasePlan) sorted by length once and cut for `world` ranks by the
    reference's rule (rank_chunks).  Every rank holds the plan of the whole database -- lengths only -- and
    materialises, interleaves and uploads just its own chunks; database indices are positions in the globally
    sorted database, whatever the number of ranks."""

    def __init__(self, plan, W: int, max_chunk_size: int, world: int, rank: int):
        self.plan, self.W = plan, W
        self.order = np.argsort(plan.lengths, kind="stable")     # reference: stable sort by length (sequences.c:125)
        self.sorted_lengths = plan.lengths[self.order]
        self.n_all = dblayout.group_lengths(self.sorted_lengths, W)
        self.mine = rank_chunks(self.n_all, W, max_chunk_size, world, rank)

    def chunk(self, k: int) -> dict:
        """Chunk k of this rank in the layout the C ABI takes: b / n / disp (reference sequences.c:479-498), its
        first sorted position s0, its number of real sequences, and the sorted sequences themselves."""
        g0, g1 = self.mine[k]
        s0, s1 = g0 * self.W, min(g1 * self.W, self.plan.nseq)
        ls = self.sorted_lengths[s0:s1]
        res = self.plan.residues_of(self.order[s0:s1])
        off = np.zeros(len(ls) + 1, np.int64)
        np.cumsum(ls, out=off[1:])
        b, n, disp = dblayout.interleave(ls, res, off, self.W)
        assert np.array_equal(n, self.n_all[g0:g1].astype(np.uint16))
        return dict(g0=g0, g1=g1, s0=s0, nseq=s1 - s0, b=b, n=n, disp=disp.astype(np.uint32), ls=ls, res=res, off=off)


def rank_step(chunks, launch, collect, nq: int, r: int, index_base: int = 0, dist=None, device=None):
    """One search step of a rank: launch(chunk) queues the search of every chunk, collect(chunk) returns its
    top list (scores [nq][r], index-in-chunk uint32 with 0xffffffff = empty); the lists are merged over the
    rank's chunks and gathered over the ranks.  Returns the global ([nq][r] scores, [nq][r] sorted positions)."""
    for c in chunks:
        launch(c)
    parts = []
    for c in chunks:
        sc, ix = collect(c)
        parts.append((sc, global_index(ix, index_base + c["s0"])))
    if not parts:  # more ranks than chunks: this rank has nothing to search
        parts = [(np.full((nq, r), -1, np.int32), np.full((nq, r), -1, np.int64))]
    sc, gix = merge_local(parts, r)
    return gather_topr(sc, gix, r, dist, device)


def global_index(chunk_index: np.ndarray, base: int) -> np.ndarray:
    """Index-in-chunk lists from oswald_hip_chunk_topr (uint32, 0xffffffff = empty slot) -> positions in the
    globally sorted database (int64, -1 = empty slot); `base` = sorted position of the chunk's first sequence."""
    ix = np.asarray(chunk_index).astype(np.int64)
    return np.where(ix == 0xFFFFFFFF, -1, ix + int(base))


def merge_local(parts, r: int):
    """Top lists of one rank's chunks ([(scores [nq][r], global index [nq][r]), ...]) -> one ([nq][r], [nq][r])."""
    if len(parts) == 1:
        return parts[0][0].astype(np.int32), parts[0][1].astype(np.int64)
    return dblayout.merge_topr_rows(np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1), r)


def gather_topr(local_scores: np.ndarray, local_global_index: np.ndarray, r: int, dist=None, device=None):
    """All-gather the per-rank top lists ([nq][r] scores, [nq][r] global indices,
    index < 0 = empty slot) and merge them on every rank.  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None or dist.get_world_size() == 1:
        return dblayout.merge_topr_rows(local_scores, local_global_index, r)
    import torch
    mine = torch.from_numpy(np.stack([local_scores.astype(np.int64), local_global_index.astype(np.int64)], axis=0))
    if device is not None:
        mine = mine.to(device)
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    allp = torch.stack(got).cpu().numpy()            # [world][2][nq][r]
    sc = np.concatenate(list(allp[:, 0]), axis=1)
    ix = np.concatenate(list(allp[:, 1]), axis=1)
    return dblayout.merge_topr_rows(sc, ix, r)
