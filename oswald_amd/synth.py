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


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Values start..start+count-1 of the SplitMix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _residue_table() -> tuple[np.ndarray, np.ndarray]:
    codes = np.array([ALPHABET.index(c) for c in _RR], dtype=np.uint8)
    p = np.array(list(_RR.values()), dtype=np.float64)
    cum = np.cumsum(p / p.sum())
    thresholds = np.minimum((cum * 2.0**32).astype(np.uint64), np.uint64(2**32 - 1))
    thresholds[-1] = np.uint64(2**32 - 1)
    return codes, thresholds


def random_residues(seed: int, start: int, count: int) -> np.ndarray:
    """`count` residue codes (uint8, 0..22) drawn with Robinson-Robinson frequencies."""
    codes, thr = _residue_table()
    u = splitmix64(seed, start, count) >> np.uint64(32)
    return codes[np.searchsorted(thr, u, side="left").clip(0, len(codes) - 1)]


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


def make_database(nseq: int, queries=None, seed: int = SEED_DB, homologs_per_query: int = 12):
    """Synthetic database: list-free representation (lengths, flat residues).

    Returns (lengths uint16 [nseq], residues uint8 [sum lengths], offsets int64
    [nseq+1]) in *generation* order (not yet length-sorted).  If `queries` is
    given, `homologs_per_query` mutated copies of each query (substitution rate
    5 %, 10 %, ... ) replace PRNG-chosen sequences.
    """
    lengths = random_lengths(seed, nseq).astype(np.int64)
    planted = {}
    if queries is not None and nseq > 0:
        ranks = splitmix64(seed ^ 0x9A17ED, 0, len(queries) * homologs_per_query)
        k = 0
        for qi, q in enumerate(queries):
            for h in range(homologs_per_query):
                idx = int(ranks[k] % np.uint64(nseq))
                while idx in planted:
                    idx = (idx + 1) % nseq
                mut = mutate(np.asarray(q, dtype=np.uint8), 0.05 * (h + 1), seed + 7919 * (qi * homologs_per_query + h + 1))
                if len(mut) > 65535:
                    mut = mut[:65535]
                planted[idx] = mut
                lengths[idx] = len(mut)
                k += 1
    offsets = np.zeros(nseq + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    residues = random_residues(seed, 0, int(offsets[-1]))
    for idx, mut in planted.items():
        residues[offsets[idx]:offsets[idx + 1]] = mut
    return lengths.astype(np.uint16), residues, offsets


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
