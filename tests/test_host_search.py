"""The product's own HOST compute path (oswald_amd/host/host_search.cpp: `oswald -m 2`, and the host share of the
hybrid mode `-m 1`) against the reference's sw_host goldens and the scalar oracle.  BASELINE configs[0]: one query of
375 residues against 1000 synthetic sequences, BLOSUM62 10/2, on the CPU.  CPU only.

This path is a mode the caller selects; the accelerator modes never fall back to it (tests/test_capi_load.py)."""
import os
import re
import subprocess

import numpy as np
import pytest

from oswald_amd import dblayout, submat, synth

import hostlib
from helpers import pack_queries

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _write_db(tmp_path, qs, nseq, **kw):
    L, R, O = synth.make_database(nseq, qs, **kw)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(nseq)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs)
    hostlib.preprocess(str(tmp_path / "db.fasta"), str(tmp_path / "db"), 2)
    return L, R, O


@pytest.mark.parametrize("v", [16, 32])
def test_c1_host_kernel_equals_reference_scores(tmp_path, v):
    """BASELINE configs[0] on both host kernels: -v 16 (SSE4.1, the path configs[0] names) and -v 32 (AVX2)."""
    g = np.load(os.path.join(GOLD, "scores.npz"))
    q1 = synth.make_queries([375])
    _write_db(tmp_path, q1, 1000, homologs_per_query=12)
    hostlib.load_queries(str(tmp_path / "q.fasta"))
    r = hostlib.assemble(str(tmp_path / "db"), 16, 134217728, 1)
    got = hostlib.host_search_chunk(0, 1, r["chunks"][0]["groups"], "blosum62", 10, 2, vector_length=v)
    np.testing.assert_array_equal(got, g["c1/scores"])
    assert got.max() > 127


@pytest.mark.parametrize("matrix,go,ge", [("pam250", 14, 2), ("blosum45", 0, 0), ("blosum62", 200, 70)])
def test_host_kernel_equals_oracle(tmp_path, oracle, matrix, go, ge):
    qs = synth.make_queries([33, 120, 260], seed=12)
    L, R, O = _write_db(tmp_path, qs, 300, seed=13, homologs_per_query=2)
    q = hostlib.load_queries(str(tmp_path / "q.fasta"))
    r = hostlib.assemble(str(tmp_path / "db"), 16, 60000, 1)
    assert r["chunk_count"] >= 2
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    for ci, c in enumerate(r["chunks"]):
        got = hostlib.host_search_chunk(ci, 3, c["groups"], matrix, go, ge, vector_length=16 if ci % 2 else 32)
        want = oracle.search_chunk_scalar(q["a"], q["m"], q["disp"][:-1].astype(np.uint32), c["b"], c["n"], c["disp"], 16, submat.load(matrix), go, ge)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("b", [0, 7, 64, 256])
def test_host_kernel_block_width(tmp_path, oracle, b):
    """`-b` (round 5): the 8-bit stage works through the query in blocks of b rows, the row above a block carried per column;
    homologs whose alignments cross the block boundaries, both kernels, every score against the scalar oracle."""
    qs = synth.make_queries([300, 77], seed=21)
    L, R, O = _write_db(tmp_path, qs, 200, seed=23, homologs_per_query=6)
    q = hostlib.load_queries(str(tmp_path / "q.fasta"))
    r = hostlib.assemble(str(tmp_path / "db"), 16, 134217728, 1)
    ch = r["chunks"][0]
    want = oracle.search_chunk_scalar(q["a"], q["m"], q["disp"][:-1].astype(np.uint32), ch["b"], ch["n"], ch["disp"], 16, submat.load("blosum62"), 10, 2)
    for v in (16, 32):
        np.testing.assert_array_equal(hostlib.host_search_chunk(0, 2, ch["groups"], "blosum62", 10, 2, vector_length=v, block_width=b), want)
    assert want.max() > 127   # (some groups went on to the int16 kernel)


def test_host_kernel_int16_ceiling(tmp_path, oracle):
    """Scores at and beyond 32767 (all-W sequences: 11 per cell): the int16 lanes saturate and are redone in int32."""
    w = synth.ALPHABET.index("W")
    q = [np.full(3100, w, np.uint8)]
    seqs = [np.full(k, w, np.uint8) for k in (5, 2977, 2978, 2979, 3100)] + [synth.random_residues(3, 0, 200)]
    synth.write_fasta(str(tmp_path / "db.fasta"), seqs)
    synth.write_fasta(str(tmp_path / "q.fasta"), q)
    hostlib.preprocess(str(tmp_path / "db.fasta"), str(tmp_path / "db"), 1)
    hostlib.load_queries(str(tmp_path / "q.fasta"))
    r = hostlib.assemble(str(tmp_path / "db"), 16, 134217728, 1)
    got = hostlib.host_search_chunk(0, 1, r["chunks"][0]["groups"], "blosum62", 10, 2)
    np.testing.assert_array_equal(hostlib.host_search_chunk(0, 1, r["chunks"][0]["groups"], "blosum62", 10, 2, vector_length=16), got)   # SSE4.1 kernel: the same ceiling handling
    assert sorted(got[0][got[0] > 0].tolist())[-4:] == [11 * 2977, 11 * 2978, 11 * 2979, 11 * 3100]


@pytest.mark.parametrize("v", [16, 32])
def test_host_kernel_int8_ceiling(tmp_path, oracle, monkeypatch, v):
    """The 8-bit first stage of the host kernels (round 5; reference HybridSearch.c:1618-1680): self-scores of 126, 127 and 128
    -- ten W (11 each) + C P / C H / C C -- sit on both sides of its ceiling; a lane at 127 sends its group to the int16 kernel,
    everything below is kept.  Both vector lengths, an odd number of groups (the AVX2 stage takes them two at a time), against the
    scalar oracle and against the same search started in int16 (OSWALD_HOST_NO_INT8, the kernel of rounds 1-4)."""
    A = synth.ALPHABET
    w, c, p_, h = A.index("W"), A.index("C"), A.index("P"), A.index("H")
    qs = [np.array([w] * 10 + tail, np.uint8) for tail in ([c, p_], [c, h], [c, c])]
    seqs = [q.copy() for q in qs] + [synth.random_residues(50 + i, 0, 12 + i % 7) for i in range(30)] + [synth.random_residues(99, 0, 300)]
    synth.write_fasta(str(tmp_path / "db.fasta"), seqs)
    synth.write_fasta(str(tmp_path / "q.fasta"), qs)
    hostlib.preprocess(str(tmp_path / "db.fasta"), str(tmp_path / "db"), 1)
    q = hostlib.load_queries(str(tmp_path / "q.fasta"))
    r = hostlib.assemble(str(tmp_path / "db"), 16, 134217728, 1)
    ch = r["chunks"][0]
    assert ch["groups"] == 3
    got = hostlib.host_search_chunk(0, 3, ch["groups"], "blosum62", 10, 2, vector_length=v)
    want = oracle.search_chunk_scalar(q["a"], q["m"], q["disp"][:-1].astype(np.uint32), ch["b"], ch["n"], ch["disp"], 16, submat.load("blosum62"), 10, 2)
    np.testing.assert_array_equal(got, want)
    assert sorted(np.unique(got.max(axis=1)).tolist()) == [126, 127, 128]
    for b in (0, 1, 5, 12, 13):   # -b: rows per block of the 8-bit stage (a 12-row query: every way to cut it)
        np.testing.assert_array_equal(hostlib.host_search_chunk(0, 3, ch["groups"], "blosum62", 10, 2, vector_length=v, block_width=b), got)
    monkeypatch.setenv("OSWALD_HOST_NO_INT8", "1")
    np.testing.assert_array_equal(hostlib.host_search_chunk(0, 3, ch["groups"], "blosum62", 10, 2, vector_length=v), got)


def test_cli_host_only_mode_report(tmp_path):
    """`oswald -O search -m 2`: BASELINE configs[0] end to end on the CPU -- the report's scores are the reference's
    (sw_host goldens), in the reference's order (descending score, ties by descending database index)."""
    g = np.load(os.path.join(GOLD, "scores.npz"))
    q1 = synth.make_queries([375])
    L, R, O = _write_db(tmp_path, q1, 1000, homologs_per_query=12)
    order = np.argsort(L.astype(np.int64), kind="stable")
    pos_of = np.empty(1000, np.int64)
    pos_of[order] = np.arange(1000)
    p = subprocess.run([hostlib.CLI, "-O", "search", "-m", "2", "-c", "8", "-v", "16", "-r", "1000", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = p.stdout
    assert "Query no.\t\t\t1\n" in out and "Query length:\t\t\t375 residues" in out and "Search speed:" in out
    rows = re.findall(r"^(\d+)\tsyn\|(\d+)\|len=(\d+)$", out, flags=re.M)
    assert len(rows) == 1000
    want = g["c1/scores"][0]
    scores = np.array([int(s) for s, _, _ in rows])
    pos = np.array([pos_of[int(i)] for _, i, _ in rows])
    np.testing.assert_array_equal(scores, want[pos])
    key = scores.astype(np.int64) * 2**32 + pos
    assert (np.diff(key) < 0).all()
    assert "Number of FPGAs:\t\t1" in out
