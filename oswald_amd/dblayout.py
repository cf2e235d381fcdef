"""Host-side data layout of the search path, mirrored from the reference.

numpy restatement of what the reference's C host does between the `.seq` file
and the device buffers (reference host/src/sequences.c:393-623,
assemble_multiple_chunks_db): sequences sorted by length, interleaved W at a
time, padded with the dummy residue 23, cut into chunks.  It feeds the C ABI
from Python (tests, bench.py); the C++ host of the command-line tool does the
same work natively (oswald_amd/host/).
"""
from __future__ import annotations

import math

import numpy as np

DUMMY = 23
FPGA_BLOCK_WIDTH = 28  # reference host/src/arguments.h:24


def sort_by_length(lengths, residues, offsets):
    """Stable ascending sort by length (reference sequences.c:125, :1140 keeps
    FASTA order among equal lengths).  Returns (order, lengths, residues, offsets)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(lengths, kind="stable")
    sl = lengths[order]
    so = np.zeros(len(sl) + 1, dtype=np.int64)
    np.cumsum(sl, out=so[1:])
    out = np.empty(int(so[-1]), dtype=np.uint8)
    # gather: source index of every residue of the sorted database
    if len(sl):
        src0 = np.repeat(np.asarray(offsets[:-1], dtype=np.int64)[order] - so[:-1], sl)
        out[:] = np.asarray(residues, dtype=np.uint8)[src0 + np.arange(int(so[-1]), dtype=np.int64)]
    return order, sl.astype(np.uint16), out, so


def group_lengths(sorted_lengths, W: int, round_to: int = FPGA_BLOCK_WIDTH) -> np.ndarray:
    """n_g = length of the longest (= last) sequence of group g, rounded up to a
    multiple of `round_to` (reference sequences.c:457-463)."""
    L = np.asarray(sorted_lengths, dtype=np.int64)
    N = len(L)
    G = (N + W - 1) // W
    last = np.minimum(np.arange(1, G + 1) * W - 1, N - 1)
    n = L[last]
    if round_to > 1:
        n = (n + round_to - 1) // round_to * round_to
    if len(n) and int(n.max()) > 65535:
        raise ValueError(f"a group of padded length {int(n.max())} does not fit the uint16 group lengths "
                         f"(sequences are limited to {65535 // round_to * round_to} residues)")
    return n


def interleave(sorted_lengths, sorted_residues, sorted_offsets, W: int = 16, round_to: int = FPGA_BLOCK_WIDTH,
               g_begin: int = 0, g_end: int | None = None):
    """Interleaved groups g_begin..g_end-1: b[disp[g] + j*W + lane] (reference
    sequences.c:479-498).  Returns (b uint8, n uint16 [groups], disp uint64 [groups])."""
    L = np.asarray(sorted_lengths, dtype=np.int64)
    N = len(L)
    n_all = group_lengths(L, W, round_to)
    G = len(n_all)
    if g_end is None:
        g_end = G
    n = n_all[g_begin:g_end]
    disp = np.zeros(len(n) + 1, dtype=np.int64)
    np.cumsum(n * W, out=disp[1:])
    b = np.full(int(disp[-1]), DUMMY, dtype=np.uint8)
    s0, s1 = g_begin * W, min(g_end * W, N)
    if s1 > s0:
        seq = np.arange(s0, s1, dtype=np.int64)
        ls = L[s0:s1]
        base = disp[(seq // W) - g_begin] + (seq % W)          # position of residue 0 of each sequence
        tot = int(ls.sum())
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(ls) - ls, ls)
        dst = np.repeat(base, ls) + within * W
        o = np.asarray(sorted_offsets, dtype=np.int64)
        b[dst] = np.asarray(sorted_residues, dtype=np.uint8)[int(o[s0]):int(o[s1])]
    return b, n.astype(np.uint16), disp[:-1].astype(np.uint64)


def chunk_plan(n, W: int, max_chunk_size: int, num_devices: int = 1):
    """Group ranges of the chunks exactly as the reference cuts them
    (sequences.c:505-541): with several devices the target is ceil(vD/ndev)
    (halved further until it fits max_chunk_size) and a chunk closes just after
    it passes the target; a single device fills up to max_chunk_size."""
    sizes = np.asarray(n, dtype=np.int64) * W
    vD = int(sizes.sum())
    if num_devices > 1:
        buffer_size = math.ceil(vD / num_devices)
        while buffer_size > max_chunk_size:
            buffer_size = math.ceil(buffer_size / num_devices)
    else:
        buffer_size = max_chunk_size
    chunks, i, G = [], 0, len(sizes)
    while i < G:
        j, chunk_size = 0, 0
        accum = int(sizes[i])
        start = i
        while i < G and chunk_size <= buffer_size and chunk_size + accum <= max_chunk_size:
            chunk_size += accum
            j += 1
            i += 1
            if i < G:
                accum = int(sizes[i])
        if j == 0:
            raise ValueError("a single group does not fit max_chunk_size")
        chunks.append((start, start + j))
    return chunks


def topr_reference_order(scores: np.ndarray, r: int):
    """Top r of a score row in the order of the reference's sort_scores()
    (utils.c:3-86): descending score, ties by DESCENDING database index."""
    scores = np.asarray(scores)
    idx = np.arange(len(scores))
    order = np.lexsort((-idx, -scores.astype(np.int64)))
    top = order[:r]
    return scores[top].astype(np.int32), top.astype(np.uint32)


def merge_topr(parts, r: int):
    """Merge per-shard (scores, global_index) top lists with the same rule; index < 0 = empty slot.
    Returns (scores int32, global index int64)."""
    sc = np.concatenate([p[0] for p in parts])
    ix = np.concatenate([p[1] for p in parts]).astype(np.int64)
    keep = ix >= 0
    sc, ix = sc[keep], ix[keep]
    order = np.lexsort((-ix, -sc.astype(np.int64)))[:r]
    return sc[order].astype(np.int32), ix[order]


def merge_topr_rows(scores: np.ndarray, index: np.ndarray, r: int):
    """merge_topr for all queries at once: scores / index [nq][K] candidates (index < 0 = empty slot) ->
    ([nq][r] int32, [nq][r] int64), empty slots (-1, -1).  The order "descending score, ties by descending
    index" is the descending order of the key score << 32 | index (scores >= 0, index < 2^32)."""
    scores = np.asarray(scores)
    index = np.asarray(index, dtype=np.int64)
    valid = index >= 0
    key = np.where(valid, (((scores.astype(np.uint64) << np.uint64(32)) | index.astype(np.uint64)) << np.uint64(1)) | np.uint64(1), np.uint64(0))
    key = np.sort(key, axis=1)[:, ::-1][:, :r]
    if key.shape[1] < r:
        key = np.concatenate([key, np.zeros((key.shape[0], r - key.shape[1]), np.uint64)], axis=1)
    have = (key & np.uint64(1)) != 0
    body = key >> np.uint64(1)
    out_s = np.where(have, (body >> np.uint64(32)).astype(np.int64), -1).astype(np.int32)
    out_i = np.where(have, (body & np.uint64(0xFFFFFFFF)).astype(np.int64), -1)
    return out_s, out_i
