"""Every GPU configuration of BASELINE.json at its full size, through the C ABI:
C2 (20 queries x 100 000 sequences, BLOSUM62 10/2), C3 (the same database,
PAM250 14/2), C5 (one 5000-residue query x the same 100 000 sequences) and the
C4 database (1 000 000 sequences, ~364.6 M residues) on one GPU, chunked by the
reference's -k rule like the command-line tool does.  Checked through
size-independent properties -- idempotence, chunking invariance, device top-r
== top-r of the downloaded table, planted homologs on top, score bounds -- and
bit for bit against the CPU port (AVX2 int8->int16->int32) on every k-th
32-lane group INCLUDING the last groups, which hold the longest sequences (the
expensive tail: widest wave geometries, workgroup items)."""
import numpy as np
import pytest

from oswald_amd import dblayout, multigpu, submat, synth

from helpers import pack_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def db100k():
    qs = synth.make_queries(synth.default_query_lengths())
    L, R, O = synth.make_database(100000, qs)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    return qs, sl, sr, so, b, n, disp.astype(np.uint32)


@pytest.fixture(scope="module")
def plan100k():
    """The same database as a plan (lengths + planted homologs and the query each is a copy of) and its sort order."""
    qs = synth.make_queries(synth.default_query_lengths())
    plan = synth.DatabasePlan(100000, qs, synth.SEED_DB, 12)
    return plan, np.argsort(plan.lengths, kind="stable")


def pin_to_oracle(toppin, oracle, plan, order, qs, sm, go, ge, sc, pos, score_at):
    """The metric's clause "top-10 score bit-exact" against the ORACLE: the scalar restatement on the nq x 10 listed
    (query, sequence) pairs and on all 12 planted homologs of every query; no planted copy that outscores a list's
    last entry is absent from it (oracle/toppin.py)."""
    pin = toppin.pin_top_list(oracle, plan, order, qs, sm, go, ge, sc, pos, score_at=score_at)
    assert pin["ok"], pin
    assert pin["list_pairs"] == len(qs) * 10 and pin["planted"] == len(qs) * 12 and pin["planted_absent"] == 0, pin


def sample_groups(nseq, W, stride, tail):
    """Every stride-th W-lane group plus the last `tail` groups -> sequence indices."""
    ng = (nseq + W - 1) // W
    groups = np.unique(np.concatenate([np.arange(0, ng, stride), np.arange(max(0, ng - tail), ng)]))
    seqs = (groups[:, None] * W + np.arange(W)[None, :]).reshape(-1)
    return seqs[seqs < nseq]


def cpu_port_scores(oracle, a, m, ad, sl, sr, so, seqs, sm, go, ge, W=32):
    lens = np.asarray(sl)[seqs].astype(np.int64)
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    idx = np.repeat(np.asarray(so)[seqs] - off[:-1], lens) + np.arange(int(off[-1]))
    bb, nn, dd = dblayout.interleave(lens, np.asarray(sr)[idx], off, W, round_to=1)
    cpu, _ = oracle.search_chunk_simd(a, m, ad, bb, nn, dd.astype(np.uint32), W, sm, go, ge)
    return cpu[:, :len(seqs)]


def check_properties(t, sc, ix, nseq, m, sl, sm):
    for q in range(t.shape[0]):
        ws, wi = dblayout.topr_reference_order(t[q, :nseq], sc.shape[1])
        np.testing.assert_array_equal(sc[q], ws)
        np.testing.assert_array_equal(ix[q], wi)
    # bounds: 0 <= score <= best possible score of the shorter partner; padding lanes are 0
    assert (t >= 0).all() and (t[:, nseq:] == 0).all()
    diag = int(sm[np.arange(23), np.arange(23)].max())
    assert (t[:, :nseq] <= diag * np.minimum(m[:, None].astype(np.int64), np.asarray(sl)[None, :nseq].astype(np.int64))).all()
    assert (np.diff(sc.astype(np.int64), axis=1) <= 0).all()


def test_c2_full_size_properties(hip_ctx, oracle, toppin, db100k, plan100k):
    qs, sl, sr, so, b, n, disp = db100k
    nseq = len(sl)
    a, m, ad = pack_queries(qs)
    sm = submat.load("blosum62")
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    h = hip_ctx.chunk_upload(b, n, disp, 16)
    t1 = np.zeros((len(qs), len(n) * 16), np.int32)
    t2 = np.zeros_like(t1)
    hip_ctx.chunk_search(h, t1)
    hip_ctx.wait()
    hip_ctx.chunk_search(h, t2)
    hip_ctx.wait()
    np.testing.assert_array_equal(t1, t2)                      # idempotent, no stale state between launches
    sc, ix = hip_ctx.chunk_topr(h, nseq, 10)
    hip_ctx.chunk_release(h)
    check_properties(t1, sc, ix, nseq, m, sl, sm)
    # the 12 planted copies of every query dominate its top-10 (well beyond the int8 range)
    assert (sc[:, 0] > 127).all() and (sc[:, 9] > 60).all()
    pin_to_oracle(toppin, oracle, *plan100k, qs, sm, 10, 2, sc, ix, lambda q, p: t1[q, p])
    seqs = sample_groups(nseq, 32, 40, 8)
    np.testing.assert_array_equal(t1[:, seqs], cpu_port_scores(oracle, a, m, ad, sl, sr, so, seqs, sm, 10, 2))
    # chunked the reference's way (8 MiB chunks) == one chunk
    plan = dblayout.chunk_plan(n, 16, 8 << 20, 1)
    assert len(plan) >= 4
    parts = []
    for g0, g1 in plan:
        pb, pn, pd = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
        out = np.zeros((len(qs), len(pn) * 16), np.int32)
        hip_ctx.search_chunk_async(pb, pn, pd.astype(np.uint32), out, 16)
        hip_ctx.wait()
        parts.append(out)
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), t1)


def test_c3_pam250_full_size(hip_ctx, oracle, toppin, db100k, plan100k):
    """BASELINE configs[2]: the same queries and database with PAM250, gap 14/2, in the named cell mode -- int8 cells
    (cell_bits = 8: SWAR 8-bit first pass, int16 re-run of what leaves its range, int32 beyond) -- and on the
    default int16 cells: both must give the reference's exact scores, i.e. identical tables."""
    qs, sl, sr, so, b, n, disp = db100k
    nseq = len(sl)
    a, m, ad = pack_queries(qs)
    sm = submat.load("pam250")
    tables = {}
    for bits in (8, 16):
        hip_ctx.set_scoring(sm, 14, 2, bits)
        hip_ctx.set_queries(a, m, ad)
        h = hip_ctx.chunk_upload(b, n, disp, 16)
        t = np.zeros((len(qs), len(n) * 16), np.int32)
        hip_ctx.chunk_search(h, t)
        hip_ctx.wait()
        sc, ix = hip_ctx.chunk_topr(h, nseq, 10)
        to16, to32 = hip_ctx.rerun_counts()
        hip_ctx.chunk_release(h)
        check_properties(t, sc, ix, nseq, m, sl, sm)
        assert (sc[:, 0] > 127).all()
        pin_to_oracle(toppin, oracle, *plan100k, qs, sm, 14, 2, sc, ix, lambda q, p: t[q, p])
        tables[bits] = t
        if bits == 8:
            assert 0 < to16 < 0.05 * t.size      # the homologs and a few strong random hits, not the bulk
    np.testing.assert_array_equal(tables[8], tables[16])
    seqs = sample_groups(nseq, 32, 40, 8)
    np.testing.assert_array_equal(tables[8][:, seqs], cpu_port_scores(oracle, a, m, ad, sl, sr, so, seqs, sm, 14, 2))


def test_c5_long_query_full_size(hip_ctx, oracle, db100k):
    """BASELINE configs[4]: one query of 5000 residues against all 100 000 sequences plus a planted 5 % copy of the
    query (score beyond the column-frame ceiling); rounds spill their boundary rows to HBM, the longest
    blocks run as workgroup items at wide geometries."""
    _, sl, sr, so, _, _, _ = db100k
    q = synth.make_queries([5000])[0]
    mut = synth.mutate(q, 0.05, 123)
    # the planted copy is the longest sequence: it sorts last
    lens = np.concatenate([sl.astype(np.int64), [len(mut)]])
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    allres = np.concatenate([sr, mut])
    nseq = len(lens)
    b, n, disp = dblayout.interleave(lens, allres, off, 16)
    a, m, ad = pack_queries([q])
    sm = submat.load("blosum62")
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    h = hip_ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
    t1 = np.zeros((1, len(n) * 16), np.int32)
    t2 = np.zeros_like(t1)
    hip_ctx.chunk_search(h, t1)
    hip_ctx.wait()
    hip_ctx.chunk_search(h, t2)
    hip_ctx.wait()
    np.testing.assert_array_equal(t1, t2)
    sc, ix = hip_ctx.chunk_topr(h, nseq, 10)
    hip_ctx.chunk_release(h)
    check_properties(t1, sc, ix, nseq, m, lens, sm)
    # (beyond the column-frame cell's ceiling; its 5004-column block runs on the plain biased cell, exact below 30576)
    assert t1[0, nseq - 1] == oracle.sw_scalar(q, mut, sm, 10, 2) and t1[0, nseq - 1] > 22256
    assert ix[0, 0] == nseq - 1
    seqs = sample_groups(nseq, 32, 25, 12)
    np.testing.assert_array_equal(t1[:, seqs], cpu_port_scores(oracle, a, m, ad, lens, allres, off, seqs, sm, 10, 2))


def test_c4_database_on_one_gpu_chunked(hip_ctx, oracle, toppin):
    """BASELINE configs[3]'s database (1 000 000 sequences) on ONE GPU, cut into 128 MiB chunks by the reference's
    rule (sequences.c:505-541, the -k default) and searched chunk after chunk like `oswald -O search` does;
    per-chunk device top-r merged to the global top-10 (positions in the globally sorted database)."""
    qs = synth.make_queries(synth.default_query_lengths())
    plan = synth.DatabasePlan(1000000, qs, synth.SEED_DB, 12)
    shard = multigpu.ShardedDatabase(plan, 16, 134217728, 1, 0, "reference")
    assert len(shard.mine) >= 3
    a, m, ad = pack_queries(qs)
    sm = submat.load("blosum62")
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    parts, best = [], None
    # the planted homologs' places in the sorted database: their scores are kept from every chunk's table for the oracle pin below
    inv = np.empty(plan.nseq, np.int64)
    inv[shard.order] = np.arange(plan.nseq)
    planted_at = sorted((int(inv[idx]), qi) for idx, qi in plan.planted_query.items())
    planted_score = {}
    for k in range(len(shard.mine)):
        c = shard.chunk(k)
        h = hip_ctx.chunk_upload(c["b"], c["n"], c["disp"], 16)
        t = np.zeros((len(qs), len(c["n"]) * 16), np.int32)
        hip_ctx.chunk_search(h, t)
        hip_ctx.wait()
        sc, ix = hip_ctx.chunk_topr(h, c["nseq"], 10)
        if k == len(shard.mine) - 1:       # the chunk with the longest sequences once more: idempotent
            t2 = np.zeros_like(t)
            hip_ctx.chunk_search(h, t2)
            hip_ctx.wait()
            np.testing.assert_array_equal(t, t2)
        hip_ctx.chunk_release(h)
        check_properties(t, sc, ix, c["nseq"], m, c["ls"], sm)
        seqs = sample_groups(c["nseq"], 32, 150, 6)
        np.testing.assert_array_equal(t[:, seqs], cpu_port_scores(oracle, a, m, ad, c["ls"], c["res"], c["off"], seqs, sm, 10, 2))
        parts.append((sc, multigpu.global_index(ix, c["s0"])))
        for pos, qi in planted_at:
            if c["s0"] <= pos < c["s0"] + c["nseq"]:
                planted_score[(qi, pos)] = int(t[qi, pos - c["s0"]])
        whole = (t[:, :c["nseq"]], c["s0"])
        # running global top-10 from the full tables, the slow way
        for q in range(len(qs)):
            ws, wi = dblayout.topr_reference_order(whole[0][q], 10)
            cur = (ws, wi.astype(np.int64) + whole[1])
            if best is None:
                best = [None] * len(qs)
            best[q] = cur if best[q] is None else dblayout.merge_topr([best[q], cur], 10)
        del t
    gs, gi = multigpu.merge_local(parts, 10)
    for q in range(len(qs)):
        np.testing.assert_array_equal(gs[q], best[q][0])
        np.testing.assert_array_equal(gi[q], best[q][1])
    # "top-10 score bit-exact" at the headline size, against the oracle: the 200 listed pairs and the 240 planted homologs
    assert len(planted_score) == len(qs) * 12
    pin_to_oracle(toppin, oracle, plan, shard.order, qs, sm, 10, 2, gs, gi, lambda q, p: planted_score[(q, p)])
    # the planted 5 % copies are on top, and their sorted positions hold sequences of about the query's length
    assert (gs[:, 0] > 300).all()
    assert (np.abs(shard.sorted_lengths[gi[:, 0]] - m.astype(np.int64)) <= 5).all()
