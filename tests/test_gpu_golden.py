"""GPU scores against the golden vectors produced by the compiled reference
itself (its SSE4.1 `sw_host`, oracle/gen_golden.py): BASELINE config 1 shape,
20 queries x 2000 sequences with PAM250 14/2, and the adversarial set around
the int8 / int16 ceilings.  No oracle in the loop here."""
import hashlib
import os

import numpy as np
import pytest

from oswald_amd import dblayout, submat, synth

from helpers import pack_queries

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def search(ctx, queries, b, n, disp, sm, go, ge):
    a, m, ad = pack_queries(queries)
    ctx.set_scoring(sm, go, ge)
    ctx.set_queries(a, m, ad)
    out = np.zeros((len(queries), len(n) * 16), np.int32)
    ctx.search_chunk_async(b, n, np.asarray(disp, np.uint32), out, 16)
    ctx.wait()
    return out


def test_c1_config_against_reference_scores(hip_ctx):
    g = np.load(os.path.join(GOLD, "scores.npz"))
    q1 = synth.make_queries([375])
    L, R, O = synth.make_database(1000, q1, homologs_per_query=12)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    assert hashlib.sha256(b.tobytes()).digest() == g["c1/b_sha256"].tobytes()
    got = search(hip_ctx, q1, b, n, disp, submat.load("blosum62"), 10, 2)
    np.testing.assert_array_equal(got, g["c1/scores"])


def test_multi_query_pam250_against_reference_scores(hip_ctx):
    g = np.load(os.path.join(GOLD, "scores.npz"))
    q2 = synth.make_queries(synth.default_query_lengths())
    L, R, O = synth.make_database(2000, q2, seed=77, homologs_per_query=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    assert hashlib.sha256(b.tobytes()).digest() == g["multi/b_sha256"].tobytes()
    got = search(hip_ctx, q2, b, n, disp, submat.load("pam250"), 14, 2)
    np.testing.assert_array_equal(got, g["multi/scores"])


def test_adversarial_against_reference_scores(hip_ctx):
    g = np.load(os.path.join(GOLD, "scores.npz"))
    w = synth.ALPHABET.index("W")
    qa = np.full(3100, w, np.uint8)
    qb = synth.make_queries([3100], seed=909)[0]
    got = search(hip_ctx, [qa, qb, qa[:12], qa[:1]], g["adv/b"], g["adv/n"], g["adv/disp"], submat.load("blosum62"), 10, 2)
    np.testing.assert_array_equal(got, g["adv/scores"])
    assert got.max() == 34100
