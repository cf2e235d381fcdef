"""Deterministic synthetic protein databases and queries.

The reference ships no sample data, so the benchmark workloads of
BASELINE.json are generated here: a length-binned database with
Robinson-Robinson residue frequencies and, for every query, a set of planted
mutated copies so that the top-10 is non-trivial and exercises the
int16 -> int32 re-run.  Everything is a pure function of the seeds; the PRNG is
SplitMix64 used as a counter-based generator (value k = mix(seed + (k+1)*gamma)),
which vectorises in numpy.

Residues are produced directly in the preprocessed alphabet of the reference
(host/src/sequences.c:165-175): A B C D E F G H I K L M N P Q R S T V W X Y Z
= 0..22, dummy = 23.
"""
from __future__ import annotations

import numpy as np

SEED_DB = 20160654
SEED_Q = 20150634

ALPHABET = "ABCDEFGHIKLMNPQRSTVWXYZ"  # code -> letter (23 symbols), dummy 23 prints as 'J'
_GAMMA = np.uint64(0x9E3779B97F4A7C15)

# length bins (lo, hi inclusive, percent)
LENGTH_BINS = ((50, 99, 12), (100, 199, 25), (200, 399, 36), (400, 799, 21), (800, 1599, 5), (1600, 3200, 1))

# Robinson & Robinson background frequencies of the 20 standard amino acids
_RR = {
    "A": 0.078, "R": 0.051, "N": 0.045, "D": 0.054, "C": 0.019, "Q": 0.043, "E": 0.063, "G": 0.074, "H": 0.022, "I": 0.051,
    "L": 0.090, "K": 0.057, "M": 0.022, "F": 0.039, "P": 0.052, "S": 0.071, "T": 0.058, "W": 0.013, "Y": 0.032, "V": 0.064,
}


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Values start..start+count-1 of the SplitMix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        return _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA)


def splitmix64_at(seed: int, positions: np.ndarray) -> np.ndarray:
    """The same stream at arbitrary positions (it is counter based): value k = mix(seed + (k+1)*gamma)."""
    with np.errstate(over="ignore"):
        k = np.asarray(positions).astype(np.uint64) + np.uint64(1)
        return _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA)


_RESIDUE_TABLE = None


def _residue_table():
    """(codes, thresholds, lut, lut_exact): residue k is drawn when the 32-bit variate u falls into
    (thresholds[k-1], thresholds[k]] (= codes[searchsorted(thresholds, u, 'left')]).  lut maps the top 16
    bits of u to that code wherever no threshold lies inside the 16-bit bucket (lut_exact)."""
    global _RESIDUE_TABLE
    if _RESIDUE_TABLE is None:
        codes = np.array([ALPHABET.index(c) for c in _RR], dtype=np.uint8)
        p = np.array(list(_RR.values()), dtype=np.float64)
        cum = np.cumsum(p / p.sum())
        thresholds = np.minimum((cum * 2.0**32).astype(np.uint64), np.uint64(2**32 - 1))
        thresholds[-1] = np.uint64(2**32 - 1)
        lo = np.arange(65536, dtype=np.uint64) << np.uint64(16)
        k_lo = np.searchsorted(thresholds, lo, side="left").clip(0, len(codes) - 1)
        k_hi = np.searchsorted(thresholds, lo + np.uint64(65535), side="left").clip(0, len(codes) - 1)
        _RESIDUE_TABLE = (codes, thresholds, codes[k_lo], k_lo == k_hi)
    return _RESIDUE_TABLE


def _residues_from_variates(z: np.ndarray) -> np.ndarray:
    codes, thr, lut, exact = _residue_table()
    u = z >> np.uint64(32)
    top = (u >> np.uint64(16)).astype(np.intp)
    out = lut[top]
    amb = np.flatnonzero(~exact[top])           # the few buckets a threshold cuts through
    if len(amb):
        out[amb] = codes[np.searchsorted(thr, u[amb], side="left").clip(0, len(codes) - 1)]
    return out


_BLOCK = 1 << 20  # variates per pass: temporaries stay cache-sized


def random_residues(seed: int, start: int, count: int) -> np.ndarray:
    """`count` residue codes (uint8, 0..22) drawn with Robinson-Robinson frequencies."""
    out = np.empty(count, dtype=np.uint8)
    for b0 in range(0, count, _BLOCK):
        nb = min(_BLOCK, count - b0)
        out[b0:b0 + nb] = _residues_from_variates(splitmix64(seed, start + b0, nb))
    return out


def random_residues_at(seed: int, positions: np.ndarray) -> np.ndarray:
    """random_residues at arbitrary stream positions."""
    positions = np.asarray(positions)
    out = np.empty(len(positions), dtype=np.uint8)
    for b0 in range(0, len(positions), _BLOCK):
        out[b0:b0 + _BLOCK] = _residues_from_variates(splitmix64_at(seed, positions[b0:b0 + _BLOCK]))
    return out


def random_lengths(seed: int, count: int) -> np.ndarray:
    """Sequence lengths from the bins of SURVEY.md section 8d (uint16)."""
    r = splitmix64(seed ^ 0x5EED1E46, 0, 2 * count)
    pick = (r[0::2] % np.uint64(100)).astype(np.int64)
    frac = r[1::2]
    out = np.empty(count, dtype=np.int64)
    acc = 0
    for lo, hi, pct in LENGTH_BINS:
        sel = (pick >= acc) & (pick < acc + pct)
        out[sel] = lo + (frac[sel] % np.uint64(hi - lo + 1)).astype(np.int64)
        acc += pct
    return out.astype(np.uint16)


def make_queries(lengths, seed: int = SEED_Q) -> list[np.ndarray]:
    """One residue array per requested length (stream positions do not overlap)."""
    out, pos = [], 0
    for m in lengths:
        out.append(random_residues(seed, pos, int(m)))
        pos += int(m)
    return out


def default_query_lengths(nq: int = 20, lo: int = 100, hi: int = 1000) -> list[int]:
    """m_k = lo + round((hi-lo)*k/(nq-1)): 100..1000, sum 11 000 for nq = 20."""
    if nq == 1:
        return [lo]
    return [lo + int(round((hi - lo) * k / (nq - 1))) for k in range(nq)]


def mutate(seq: np.ndarray, rate: float, seed: int) -> np.ndarray:
    """Copy of `seq` with i.i.d. substitutions at `rate` and one 1..5 residue indel."""
    n = len(seq)
    r = splitmix64(seed, 0, 2 * n + 8)
    out = seq.copy()
    hit = (r[:n] >> np.uint64(11)).astype(np.float64) / 2.0**53 < rate
    repl = random_residues(seed ^ 0xABCDEF, 0, n)
    out[hit] = repl[hit]
    ilen = 1 + int(r[2 * n] % np.uint64(5))
    pos = int(r[2 * n + 1] % np.uint64(max(1, n - ilen)))
    if int(r[2 * n + 2] & np.uint64(1)):  # deletion
        out = np.concatenate([out[:pos], out[pos + ilen:]])
    else:  # insertion
        ins = random_residues(seed ^ 0x1234567, 0, ilen)
        out = np.concatenate([out[:pos], ins, out[pos:]])
    return out.astype(np.uint8)


class DatabasePlan:
    """Everything about a synthetic database except its residues: lengths, offsets into the residue
    stream and the planted homologs.  Cheap (no per-residue work), so every rank of a sharded search
    can hold the plan of the whole database and materialise only the sequences of its own shard."""

    def __init__(self, nseq, queries, seed, homologs_per_query, rates=None):
        """rates: substitution rates of a query's planted copies, cycled through (default 5 %, 10 %, ... by copy number)"""
        self.nseq, self.seed = nseq, seed
        lengths = random_lengths(seed, nseq).astype(np.int64)
        self.planted = {}
        self.planted_query = {}   # generation-order id of a planted copy -> the query it is a copy of
        if queries is not None and nseq > 0:
            ranks = splitmix64(seed ^ 0x9A17ED, 0, len(queries) * homologs_per_query)
            k = 0
            for qi, q in enumerate(queries):
                for h in range(homologs_per_query):
                    idx = int(ranks[k] % np.uint64(nseq))
                    while idx in self.planted:
                        idx = (idx + 1) % nseq
                    rate = 0.05 * (h + 1) if rates is None else rates[h % len(rates)]
                    mut = mutate(np.asarray(q, dtype=np.uint8), rate, seed + 7919 * (qi * homologs_per_query + h + 1))
                    if len(mut) > 65520:
                        mut = mut[:65520]
                    self.planted[idx] = mut
                    self.planted_query[idx] = qi
                    lengths[idx] = len(mut)
                    k += 1
        self.lengths = lengths
        self.offsets = np.zeros(nseq + 1, dtype=np.int64)
        np.cumsum(lengths, out=self.offsets[1:])

    def residues_of(self, seq_ids) -> np.ndarray:
        """Residues of the given sequences (generation-order ids), concatenated in the order given."""
        ids = np.asarray(seq_ids, dtype=np.int64)
        ls = self.lengths[ids]
        out_off = np.zeros(len(ids) + 1, dtype=np.int64)
        np.cumsum(ls, out=out_off[1:])
        out = np.empty(int(out_off[-1]), dtype=np.uint8)
        # in slabs of sequences so that the position arrays stay small
        step = 4096
        for s0 in range(0, len(ids), step):
            s1 = min(len(ids), s0 + step)
            l = ls[s0:s1]
            tot = int(out_off[s1] - out_off[s0])
            pos = np.repeat(self.offsets[ids[s0:s1]] - (out_off[s0:s1] - out_off[s0]), l) + np.arange(tot, dtype=np.int64)
            out[out_off[s0]:out_off[s1]] = random_residues_at(self.seed, pos)
        for k in np.flatnonzero(np.isin(ids, np.fromiter(self.planted.keys(), dtype=np.int64, count=len(self.planted)))) if self.planted else ():
            out[out_off[k]:out_off[k + 1]] = self.planted[int(ids[k])]
        return out


def make_database(nseq: int, queries=None, seed: int = SEED_DB, homologs_per_query: int = 12):
    """Synthetic database: list-free representation (lengths, flat residues).

    Returns (lengths uint16 [nseq], residues uint8 [sum lengths], offsets int64
    [nseq+1]) in *generation* order (not yet length-sorted).  If `queries` is
    given, `homologs_per_query` mutated copies of each query (substitution rate
    5 %, 10 %, ... ) replace PRNG-chosen sequences.
    """
    plan = DatabasePlan(nseq, queries, seed, homologs_per_query)
    residues = random_residues(seed, 0, int(plan.offsets[-1]))
    for idx, mut in plan.planted.items():
        residues[plan.offsets[idx]:plan.offsets[idx + 1]] = mut
    return plan.lengths.astype(np.uint16), residues, plan.offsets


def to_letters(codes: np.ndarray) -> str:
    """Preprocessed codes back to FASTA letters (dummy 23 -> 'J')."""
    lut = np.frombuffer((ALPHABET + "J" * 233).encode(), dtype=np.uint8)
    return lut[np.asarray(codes, dtype=np.uint8)].tobytes().decode()


def write_fasta(path: str, seqs, titles=None, width: int = 60) -> None:
    """FASTA with a trailing newline and upper-case letters only (the reference's
    parser needs both)."""
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            t = titles[i] if titles is not None else f"syn|{i}|len={len(s)}"
            f.write(f">{t}\n")
            letters = to_letters(s)
            for k in range(0, len(letters), width):
                f.write(letters[k:k + width] + "\n")
