"""Sharding of the search path over GPUs: one process per GPU, every rank owns
a shard of the length-sorted database; the only exchange is the gather of the
per-rank top-r lists, which is what runs over RCCL/xGMI (backend "nccl") or
gloo on CPUs.  Two shard rules:

  "deal"       (default of bench.py) the database is dealt to the ranks in wave
               blocks of 128 consecutive sorted sequences -- the unit one wave
               of the search kernels works on -- in rounds of `world` blocks,
               forwards in even rounds and backwards in odd ones (a rank that
               got the shortest block of one round gets the longest of the
               next).  Every rank then sees the whole length
               distribution (same plan quality, same share of the longest
               sequences), so the ranks finish together; the reference's rule
               puts all the longest sequences into the last shard.
  "reference"  contiguous shards of equal padded size: the reference gives chunk
               c of a round to device c mod ndev (host/src/FPGAsearch.c:132-138)
               after cutting the database into chunks of ceil(vD/ndev) padded
               residues (host/src/sequences.c:510-515), and merges by memcpy
               (FPGAsearch.c:236-237).  The CLI's -f uses this rule.

Plumbing only: torch.distributed carries (score, global index) pairs; the
ordering rule is the reference's (descending score, ties by descending index,
host/src/utils.c:3-86)."""
from __future__ import annotations

import os

import numpy as np

from . import dblayout


def init_collective(backend: str, device=None):
    """torch.distributed bring-up of a rank (RANK / WORLD_SIZE / MASTER_* from the environment): "nccl" = RCCL, one
    rank per GPU; "gloo" = the rehearsal mode the caller asked for by name (ranks sharing a GPU, or no GPU at all).
    There is NO fallback from one to the other: a failed RCCL bring-up ends the rank with a non-zero exit code and the
    reason (the launcher then takes the other ranks down), so a scaling line can never quietly stop being RCCL
    evidence, and no rank can end up in a rendezvous the others are not in.  Returns torch.distributed."""
    import torch.distributed as dist
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"unknown collective backend {backend!r} (nccl = RCCL, or gloo)")
    try:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
    except Exception as e:  # noqa: BLE001 -- whatever the bring-up raised: say so and leave; never switch backends
        raise SystemExit(f"collective bring-up failed on rank {os.environ.get('RANK', '?')} with backend {backend} "
                         f"({type(e).__name__}: {str(e)[:300]}); no other backend is tried -- ask for one by name "
                         "(OSWALD_BENCH_BACKEND=gloo) if a rehearsal without RCCL is what you want")
    return dist


def rank_chunks(n_groups_len, W: int, max_chunk_size: int, world: int, rank: int):
    """Group ranges this rank searches: the database is cut by the reference's
    rule (dblayout.chunk_plan, host/src/sequences.c:505-541) and chunk c goes to
    device c mod world (host/src/FPGAsearch.c:132-138)."""
    plan = dblayout.chunk_plan(n_groups_len, W, max_chunk_size, world)
    return [p for c, p in enumerate(plan) if c % world == rank]


DEAL_BLOCK_SEQS = 128   # sequences per dealt unit = one wave block of the search kernels (sw_kernels.h OSW_BLOCK_SEQS)


def dealt_positions(nseq: int, world: int, rank: int, block_seqs: int = DEAL_BLOCK_SEQS) -> np.ndarray:
    """Sorted positions of the sequences rank `rank` owns under the "deal" rule.  Blocks are counted from the LONGEST
    end -- block u holds the sorted positions [nseq - (u+1)*128, nseq - u*128) -- so that both the one incomplete
    block and the incomplete last round hold the shortest sequences; of the blocks of round t a rank gets number
    `rank` when t is even and number world - 1 - rank when t is odd.  Returned in ascending order."""
    nblk = (nseq + block_seqs - 1) // block_seqs
    t = np.arange((nblk + world - 1) // world, dtype=np.int64)
    u = t * world + np.where(t % 2 == 0, rank, world - 1 - rank)
    u = u[u < nblk][::-1]
    pos = (nseq - (u[:, None] + 1) * block_seqs + np.arange(block_seqs, dtype=np.int64)[None, :]).reshape(-1)
    return pos[pos >= 0]


class ShardedDatabase:
    """A synthetic database (synth.DatabasePlan) sorted by length once and divided among `world` ranks by `rule`
    (module docstring).  Every rank holds the plan of the whole database -- lengths only -- and materialises,
    interleaves and uploads just its own chunks; database indices are positions in the globally sorted
    database, whatever the number of ranks and the rule (chunk()["gpos"] maps a chunk's sequences to them)."""

    def __init__(self, plan, W: int, max_chunk_size: int, world: int, rank: int, rule: str = "deal"):
        if rule not in ("deal", "reference"):
            raise ValueError(f"unknown shard rule {rule!r}")
        self.plan, self.W, self.rule = plan, W, rule
        self.order = np.argsort(plan.lengths, kind="stable")     # reference: stable sort by length (sequences.c:125)
        self.sorted_lengths = plan.lengths[self.order]
        self.n_all = dblayout.group_lengths(self.sorted_lengths, W)
        if rule == "reference":
            self.mine = rank_chunks(self.n_all, W, max_chunk_size, world, rank)
            self.pos = None
        else:
            # the rank's sequences, still sorted by length, interleaved like a database of their own; cut into as few
            # equal chunks as max_chunk_size allows
            self.pos = dealt_positions(plan.nseq, world, rank)
            n_sub = dblayout.group_lengths(self.sorted_lengths[self.pos], W) if len(self.pos) else np.zeros(0, np.int64)
            parts = max(1, -(-int(n_sub.sum()) * W // max_chunk_size))
            self.mine = dblayout.chunk_plan(n_sub, W, max_chunk_size, parts) if len(n_sub) else []

    def chunk(self, k: int) -> dict:
        """Chunk k of this rank in the layout the C ABI takes: b / n / disp (reference sequences.c:479-498), the
        sorted positions gpos of its sequences (s0 = the first), its number of real sequences, and the sorted
        sequences themselves."""
        g0, g1 = self.mine[k]
        if self.pos is None:
            s0, s1 = g0 * self.W, min(g1 * self.W, self.plan.nseq)
            gpos = np.arange(s0, s1, dtype=np.int64)
        else:
            gpos = self.pos[g0 * self.W:min(g1 * self.W, len(self.pos))]
        ls = self.sorted_lengths[gpos]
        res = self.plan.residues_of(self.order[gpos])
        off = np.zeros(len(ls) + 1, np.int64)
        np.cumsum(ls, out=off[1:])
        b, n, disp = dblayout.interleave(ls, res, off, self.W)
        if self.pos is None:
            assert np.array_equal(n, self.n_all[g0:g1].astype(np.uint16))
        return dict(g0=g0, g1=g1, s0=int(gpos[0]) if len(gpos) else 0, gpos=gpos, nseq=len(gpos), b=b, n=n, disp=disp.astype(np.uint32),
                    ls=ls, res=res, off=off)


def rank_step(chunks, launch, collect, nq: int, r: int, index_base: int = 0, dist=None, device=None, collect_rank=None):
    """One search step of a rank: launch(chunk) queues the search of every chunk; then either collect_rank() returns
    the rank's merged top list (oswald_hip_topr: [nq][r] scores, [nq][r] uint32 database indices, 0xffffffff = empty
    slot), or collect(chunk) returns each chunk's list (scores [nq][r], index-in-chunk uint32 with 0xffffffff = empty)
    and the lists are merged here.  The rank lists are gathered over the ranks.  Returns the global ([nq][r] scores,
    [nq][r] sorted positions)."""
    for c in chunks:
        launch(c)
    if collect_rank is not None:
        sc, ix = collect_rank()
        ix = np.asarray(ix).astype(np.int64)
        sc, gix = np.asarray(sc, dtype=np.int32), np.where(ix == 0xFFFFFFFF, -1, ix + int(index_base))
    else:
        parts = []
        for c in chunks:
            sc, ix = collect(c)
            parts.append((sc, global_index(ix, index_base + c["s0"], c.get("gpos"), index_base)))
        if not parts:  # more ranks than chunks: this rank has nothing to search
            parts = [(np.full((nq, r), -1, np.int32), np.full((nq, r), -1, np.int64))]
        sc, gix = merge_local(parts, r)
    return gather_topr(sc, gix, r, dist, device)


def global_index(chunk_index: np.ndarray, base: int, gpos=None, gpos_base: int = 0) -> np.ndarray:
    """Index-in-chunk lists from oswald_hip_chunk_topr (uint32, 0xffffffff = empty slot) -> positions in the
    globally sorted database (int64, -1 = empty slot): gpos[index] + gpos_base when the chunk carries the sorted
    positions of its sequences (a dealt chunk is not one contiguous run), else index + base (`base` = sorted
    position of the chunk's first sequence)."""
    ix = np.asarray(chunk_index).astype(np.int64)
    empty = ix == 0xFFFFFFFF
    if gpos is not None:
        g = np.asarray(gpos, dtype=np.int64)
        return np.where(empty, -1, g[np.where(empty, 0, ix).clip(0, max(len(g) - 1, 0))] + int(gpos_base)) if len(g) else np.full(ix.shape, -1, np.int64)
    return np.where(empty, -1, ix + int(base))


def merge_rows(scores: np.ndarray, index: np.ndarray, r: int):
    """[nq][K] candidates (index < 0 = empty slot) -> ([nq][r] int32, [nq][r] int64), empty slots (-1, -1): the
    library's one merge implementation (oswald_hip_merge_candidates: descending score, ties by descending index)."""
    from . import capi
    index = np.asarray(index, dtype=np.int64)
    cs = np.where(index < 0, -1, np.asarray(scores)).astype(np.int32)
    out_s, out_i = capi.merge_candidates(cs, np.where(index < 0, 0, index).astype(np.uint32), r)
    return out_s, np.where(out_s < 0, -1, out_i.astype(np.int64))


def merge_local(parts, r: int):
    """Top lists of one rank's chunks ([(scores [nq][r], global index [nq][r]), ...]) -> one ([nq][r], [nq][r])."""
    if len(parts) == 1:
        return parts[0][0].astype(np.int32), parts[0][1].astype(np.int64)
    return merge_rows(np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1), r)


def gather_topr(local_scores: np.ndarray, local_global_index: np.ndarray, r: int, dist=None, device=None):
    """All-gather the per-rank top lists ([nq][r] scores, [nq][r] global indices,
    index < 0 = empty slot) and merge them on every rank.  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None or dist.get_world_size() == 1:
        return merge_rows(local_scores, local_global_index, r)
    import torch
    mine = torch.from_numpy(np.stack([local_scores.astype(np.int64), local_global_index.astype(np.int64)], axis=0))
    if device is not None:
        mine = mine.to(device)
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    allp = torch.stack(got).cpu().numpy()            # [world][2][nq][r]
    sc = np.concatenate(list(allp[:, 0]), axis=1)
    ix = np.concatenate(list(allp[:, 1]), axis=1)
    return merge_rows(sc, ix, r)
