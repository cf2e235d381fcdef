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


def gather_topr(local_scores: np.ndarray, local_global_index: np.ndarray, r: int, dist=None, device=None):
    """All-gather the per-rank top lists ([nq][r] scores, [nq][r] global indices,
    index < 0 = empty slot) and merge them on every rank.  `dist` is
    torch.distributed (initialised) or None for a single process."""
    nq = local_scores.shape[0]
    if dist is None or dist.get_world_size() == 1:
        parts = [(local_scores, local_global_index)]
    else:
        import torch
        mine = torch.from_numpy(np.stack([local_scores.astype(np.int64), local_global_index.astype(np.int64)], axis=0))
        if device is not None:
            mine = mine.to(device)
        got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(got, mine)
        allp = torch.stack(got).cpu().numpy()
        parts = [(allp[k, 0], allp[k, 1]) for k in range(allp.shape[0])]
    out_s = np.empty((nq, r), np.int32)
    out_i = np.empty((nq, r), np.int64)
    out_s.fill(-1)
    out_i.fill(-1)
    for q in range(nq):
        s_, i_ = dblayout.merge_topr([(p[0][q].astype(np.int32), p[1][q]) for p in parts], r)
        out_s[q, :len(s_)] = s_
        out_i[q, :len(i_)] = i_
    return out_s, out_i
