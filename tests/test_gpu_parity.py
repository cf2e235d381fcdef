"""GPU parity: the HIP path through the C ABI against the CPU oracle.

Bar: bit-exact int32 scores for every (query, database sequence), i.e. exactly
what the reference's host path reports after its int8->int16->int32 escalation
(reference host/src/HybridSearch.c:1573-1880).
"""
import numpy as np
import pytest

from oswald_amd import dblayout, submat, synth

from helpers import db_from_sequences, layout, pack_queries, random_db

pytestmark = pytest.mark.gpu


def run_gpu(ctx, queries, b, n, disp, W, sm, go, ge, cell_bits=0, resident=False):
    a, m, ad = pack_queries(queries)
    ctx.set_scoring(sm, go, ge, cell_bits)
    ctx.set_queries(a, m, ad)
    out = np.full((len(queries), len(n) * W), -7, dtype=np.int32)
    if resident:
        h = ctx.chunk_upload(b, n, disp, W)
        ctx.chunk_search(h, out)
        ctx.wait()
        ctx.chunk_release(h)
    else:
        ctx.search_chunk_async(b, n, disp, out, W)
        ctx.wait()
    return out


def expect(oracle, queries, b, n, disp, W, sm, go, ge):
    a, m, ad = pack_queries(queries)
    return oracle.search_chunk_scalar(a, m, ad, b, n, disp, W, sm, go, ge)


@pytest.mark.parametrize("nseq", [1, 15, 16, 17, 127, 128, 129, 300])
def test_small_databases_all_group_shapes(hip_ctx, oracle, nseq):
    qs = synth.make_queries([1, 3, 4, 5, 31, 32, 33, 64, 97, 100], seed=11)
    L, R, O = random_db(nseq, seed=nseq, max_len=120, queries=qs[-3:], homologs=1)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("matrix,go,ge", [("blosum62", 10, 2), ("pam250", 14, 2), ("blosum45", 0, 0), ("pam30", 30, 5), ("blosum90", 3, 1),
                                          ("blosum62", 200, 70), ("blosum62", 3, 60), ("blosum62", 0, 9), ("blosum62", 1000, 1000),
                                          # gap open around 1024: beyond it the column-frame cell's 32-bit subtract of the penalty could borrow
                                          # across the halves of the packed pair, so the kernel must take the plain cell (osw_frame_cell_takes)
                                          ("blosum62", 1023, 1), ("blosum62", 1024, 0), ("blosum62", 1025, 1), ("pam250", 2000, 2)])
def test_matrices_and_gaps(hip_ctx, oracle, matrix, go, ge):
    qs = synth.make_queries([50, 129, 375], seed=5)
    L, R, O = random_db(200, seed=77, max_len=400, queries=qs, homologs=3)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load(matrix)
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge, resident=True)
    want = expect(oracle, qs, b, n, disp, 16, sm, go, ge)
    np.testing.assert_array_equal(got, want)
    assert want.max() > 127  # the planted copies are beyond the int8 range


def test_lane_width_32_layout(hip_ctx, oracle):
    qs = synth.make_queries([40, 200], seed=8)
    L, R, O = random_db(150, seed=9, max_len=150, queries=qs, homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 32, round_to=1)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 32, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 32, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


def test_int16_ceiling_forces_exact_int32_rerun(hip_ctx, oracle):
    """Scores around and beyond 32767: all-W sequences score 11 per cell on the
    diagonal with BLOSUM62; 2978 W's = 32758, 2979 = 32769."""
    w = synth.ALPHABET.index("W")
    q = np.full(3100, w, dtype=np.uint8)
    seqs = [np.full(k, w, dtype=np.uint8) for k in (1, 11, 2977, 2978, 2979, 2980, 3100)]
    seqs += [synth.random_residues(3, 0, 500), synth.mutate(q, 0.05, 4)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, sl, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, [q, q[:700]], b, n, disp, 16, sm, 10, 2, resident=True)
    want = expect(oracle, [q, q[:700]], b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want.max() == 11 * 3100 and (want == 32758).any() and (want == 32769).any()
    _, _, rerun = hip_ctx.kernel_stats()
    assert rerun >= 1


def test_exact_score_32767_is_rerun_and_kept(hip_ctx, oracle):
    """A true score of exactly 32767 is indistinguishable from saturation in the
    int16 pass; like the reference (== 32767 test, HybridSearch.c:1774) it is
    re-run and must come back as 32767."""
    sm = submat.load("blosum62").copy()
    w = synth.ALPHABET.index("W")
    sm[w, w] = 127   # 258 * 127 = 32766, +1 via a custom cell
    c = synth.ALPHABET.index("C")
    sm[c, c] = 1
    q = np.concatenate([np.full(258, w, np.uint8), np.full(1, c, np.uint8)])
    L, R, O = db_from_sequences([q.copy(), q[:258].copy(), q[:100].copy()])
    b, n, disp, _, _ = layout(L, R, O, 16)
    got = run_gpu(hip_ctx, [q], b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, [q], b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want.max() == 32767


def test_int32_mode_equals_int16_mode(hip_ctx, oracle):
    qs = synth.make_queries([17, 260], seed=21)
    L, R, O = random_db(140, seed=31, max_len=300, queries=qs, homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("pam250")
    got32 = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 14, 2, cell_bits=32)
    got16 = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 14, 2, cell_bits=16)
    want = expect(oracle, qs, b, n, disp, 16, sm, 14, 2)
    np.testing.assert_array_equal(got32, want)
    np.testing.assert_array_equal(got16, want)


def _self_scoring(target, seed):
    """A sequence whose BLOSUM62 self-alignment scores exactly `target`: W (11), A (4) and E (5) in a seeded order."""
    rng = np.random.default_rng(seed)
    w = (target - 40) // 11
    rest = target - 11 * w           # 40..50: always 4a + 5b
    b5 = next(k for k in range(11) if (rest - 5 * k) % 4 == 0 and rest - 5 * k >= 0)
    a4 = (rest - 5 * b5) // 4
    seq = np.array([19] * w + [0] * a4 + [4] * b5, dtype=np.uint8)   # codes (synth.ALPHABET): A=0 E=4 W=19
    rng.shuffle(seq)
    assert 11 * w + 4 * a4 + 5 * b5 == target
    return seq


def test_scores_around_2048(hip_ctx, oracle, cell_bits=16):
    """Self-alignments scoring 2040..2056 and 4000 (integers from 2048 on are no longer exact in fp16: the int16 cell
    only uses the fp16 *ordering* of its bit patterns, so nothing may change here)."""
    targets = [2040, 2045, 2046, 2047, 2048, 2049, 2050, 2056, 4000]
    qs = [_self_scoring(t, 100 + t) for t in targets]
    seqs = [q.copy() for q in qs] + [synth.mutate(q, 0.02, 7 + i) for i, q in enumerate(qs)]
    seqs += [synth.random_residues(900 + i, 0, 150 + 7 * i) for i in range(40)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2, cell_bits=cell_bits)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    best = want.max(axis=1)
    assert best.tolist() == targets
    assert ((want >= 2040) & (want < 2048)).any() and (want >= 2048).any()


def test_long_query_many_strips(hip_ctx, oracle):
    """m = 2500: 79 strips, every one spilling its bottom row through HBM."""
    q = synth.make_queries([2500], seed=99)[0]
    L, R, O = random_db(130, seed=41, min_len=20, max_len=700, queries=[q], homologs=3)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, [q], b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, [q], b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


def test_dummy_residues_and_odd_codes(hip_ctx, oracle):
    """J/O/U map to code 23 (reference sequences.c:167-169) in queries and in the
    database; 23 scores 0 everywhere (submat.c row/col 23)."""
    rng = np.random.default_rng(5)
    qs = [rng.integers(0, 24, 90).astype(np.uint8), np.full(10, 23, np.uint8)]
    seqs = [rng.integers(0, 24, int(l)).astype(np.uint8) for l in rng.integers(1, 90, 40)]
    seqs.append(np.full(30, 23, np.uint8))
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


def test_topr_on_device_tie_order(hip_ctx, oracle):
    qs = synth.make_queries([30, 60], seed=2)
    L, R, O = random_db(500, seed=6, max_len=60)
    b, n, disp, sl, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    h = hip_ctx.chunk_upload(b, n, disp, 16)
    out = np.zeros((2, len(n) * 16), np.int32)
    hip_ctx.chunk_search(h, out)
    hip_ctx.wait()
    sc, ix = hip_ctx.chunk_topr(h, 500, 25)
    hip_ctx.chunk_release(h)
    for q in range(2):
        wsc, wix = dblayout.topr_reference_order(out[q, :500], 25)
        np.testing.assert_array_equal(sc[q], wsc)
        np.testing.assert_array_equal(ix[q], wix)
        osc, oix = oracle.sort_scores(out[q, :500])
        np.testing.assert_array_equal(sc[q], osc[:25])
        np.testing.assert_array_equal(ix[q], oix[:25])
    assert len(np.unique(out[0, :500])) < 400  # ties are present


def test_chunks_equal_single_chunk(hip_ctx, oracle):
    """Cutting the database into chunks the reference's way (sequences.c:505-541)
    and searching them one by one gives the same score table."""
    qs = synth.make_queries([80, 33], seed=14)
    L, R, O = random_db(700, seed=15, max_len=180)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    sm = submat.load("blosum62")
    whole = run_gpu(hip_ctx, qs, bfull, nfull, dfull.astype(np.uint32), 16, sm, 10, 2)
    plan = dblayout.chunk_plan(nfull, 16, 30000, 1)
    assert len(plan) >= 2
    parts = []
    for g0, g1 in plan:
        b, n, disp = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
        parts.append(run_gpu(hip_ctx, qs, b, n, disp.astype(np.uint32), 16, sm, 10, 2))
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), whole)


@pytest.mark.parametrize("lg", [0, 1, 2, 3, 4, 5])
def test_every_wave_geometry(hip_ctx, oracle, lg, monkeypatch):
    """G = 2^lg lane groups per wave (systolic strips inside the wave): the
    result must not depend on the geometry."""
    monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    qs = synth.make_queries([3, 8, 37, 130, 301], seed=40 + lg)
    L, R, O = random_db(260, seed=50 + lg, max_len=90, queries=qs[-2:], homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lg", [0, 3, 6])
def test_every_wave_geometry_int32(hip_ctx, oracle, lg, monkeypatch):
    monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    qs = synth.make_queries([5, 70, 260], seed=60 + lg)
    L, R, O = random_db(140, seed=70 + lg, max_len=80, queries=qs[-1:], homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("pam250")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 14, 2, cell_bits=32)
    want = expect(oracle, qs, b, n, disp, 16, sm, 14, 2)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lg", [2, 4, 6])
def test_workgroup_items(hip_ctx, oracle, lg, monkeypatch):
    """Heavy-item path: four waves of a workgroup on four sub-blocks of one item,
    sharing the profile slice (forced here on a small input)."""
    monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    monkeypatch.setenv("OSWALD_HIP_FORCE_WG", "1")
    qs = synth.make_queries([9, 64, 700, 1500], seed=80 + lg)
    L, R, O = random_db(200, seed=90 + lg, max_len=150, queries=qs[-2:], homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


def test_edge_cases_empty_inputs(hip_ctx, oracle):
    """Empty query (length 0) scores 0 everywhere; an empty chunk is a no-op;
    errors are reported, not swallowed."""
    from oswald_amd import capi
    sm = submat.load("blosum62")
    qs = [np.zeros(0, np.uint8), synth.make_queries([20], seed=1)[0]]
    L, R, O = random_db(40, seed=2, max_len=50)
    b, n, disp, _, _ = layout(L, R, O, 16)
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert (got[0] == 0).all()
    # empty chunk
    a, m, ad = pack_queries(qs[1:])
    hip_ctx.set_queries(a, m, ad)
    h = hip_ctx.chunk_upload(np.zeros(0, np.uint8), np.zeros(0, np.uint16), np.zeros(0, np.uint32), 16)
    hip_ctx.chunk_search(h, None)
    hip_ctx.wait()
    hip_ctx.chunk_release(h)
    # a group running past the chunk buffer is rejected before anything is launched
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_upload(np.zeros(100, np.uint8), np.array([28], np.uint16), np.array([0], np.uint32), 16)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_upload(b, n, disp, 24)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_topr(9999, 1, 1)


def test_max_length_sequence_and_many_queries(hip_ctx, oracle):
    """One very long database sequence (8 000 residues) among short ones and 40
    queries: item planning with extreme cost ratios."""
    rng = np.random.default_rng(8)
    qs = [rng.integers(0, 23, int(l)).astype(np.uint8) for l in rng.integers(5, 60, 40)]
    long_seq = synth.random_residues(77, 0, 8000)
    long_seq[4000:4050] = qs[-1][:50] if len(qs[-1]) >= 50 else long_seq[4000:4050]
    seqs = [synth.random_residues(100 + i, 0, int(l)) for i, l in enumerate(rng.integers(1, 40, 30))] + [long_seq]
    from helpers import db_from_sequences
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum80")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 11, 1)
    want = expect(oracle, qs, b, n, disp, 16, sm, 11, 1)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lg,wg", [(-1, -1), (0, 0), (2, 0), (4, 0), (2, 1), (4, 1), (6, 1)])
def test_query_pairs(hip_ctx, oracle, lg, wg, monkeypatch):
    """Query batching: two queries share a lane (osw_sw_pk16q).  OSWALD_HIP_PAIRS=2 pairs every
    neighbour in length order, however different the lengths, at every geometry."""
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    if lg >= 0:
        monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    if wg >= 0:
        monkeypatch.setenv("OSWALD_HIP_FORCE_WG", str(wg))
    qs = synth.make_queries([1, 7, 64, 65, 130, 131, 300, 1200, 40], seed=300 + lg)   # 4 pairs + one single
    L, R, O = random_db(270, seed=310 + lg, max_len=110, queries=qs[4:8], homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lg,wg", [(-1, -1), (0, 0), (1, 0), (3, 0), (4, 0), (2, 1), (4, 1), (6, 1)])
@pytest.mark.parametrize("go,ge", [(10, 2), (3, 0)])
def test_pair_tails(hip_ctx, oracle, lg, wg, go, ge, monkeypatch):
    """Tails: a pair item marked SHORT runs the shorter query's rows as a pair -- its last strips end on a row of the longer query,
    which row depends on the geometry -- and then, in the same wave at the same geometry, the rest of the longer query on the
    single-query cell, starting from the bottom rows the two passes left (pass 0 in the wave's hand region, pass 1 in place).
    Forced on (OSWALD_HIP_PAIR_TAILS=2: every eligible item, whatever the cost model says), at every geometry of the items, wave and
    workgroup items, for pairs whose lengths differ by less than a strip, by several strips (a tail of several rounds) and not at
    all, with homologs of the longer queries (alignments that cross the hand-over row) and with gap extend 0 (the plain cell)."""
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "2")
    if lg >= 0:
        monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    if wg >= 0:
        monkeypatch.setenv("OSWALD_HIP_FORCE_WG", str(wg))
    qs = synth.make_queries([5, 40, 64, 111, 200, 207, 300, 300, 350, 900], seed=500 + lg)   # five pairs, no single query (tails are planned for such sets)
    L, R, O = random_db(300, seed=510 + lg, max_len=140, queries=[qs[3], qs[5], qs[9], qs[9][500:]], homologs=3)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge)
    want = expect(oracle, qs, b, n, disp, 16, sm, go, ge)
    np.testing.assert_array_equal(got, want)
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "1")            # ... the cost model's choice, item by item
    np.testing.assert_array_equal(run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge), want)
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "0")            # ... and the same table with padded pairs
    np.testing.assert_array_equal(run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge), want)
    # a set with a single query beside its pairs keeps its padded pairs (the hand regions are that launch's half of the scratch)
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "2")
    qs1 = qs + synth.make_queries([33], seed=77)
    np.testing.assert_array_equal(run_gpu(hip_ctx, qs1, b, n, disp, 16, sm, go, ge), expect(oracle, qs1, b, n, disp, 16, sm, go, ge))


def test_pair_tails_over_the_int16_ceiling(hip_ctx, oracle, monkeypatch):
    """The longer query of a pair reaches the int16 cells' ceiling in its TAIL rows only (all-W: 11 per cell; 2 000 rows of the pair
    + 1 000 of the tail): the tail queues the sequence for the int32 re-run like any other item."""
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "2")
    w = synth.ALPHABET.index("W")
    qa, qb = np.full(2000, w, np.uint8), np.full(3000, w, np.uint8)
    seqs = [np.full(k, w, np.uint8) for k in (5, 1990, 2500, 3000)] + [synth.random_residues(3, 0, 300)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, [qa, qb], b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, [qa, qb], b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want[1].max() == 33000 and hip_ctx.rerun_counts()[1] >= 1


@pytest.mark.parametrize("lg", [-1, 0, 2])
def test_pair_tails_at_the_frame_limit(hip_ctx, oracle, lg, monkeypatch):
    """blocks whose column count lies at the column-frame cell's limit ((columns + 2 G + 2) x ge <= 8192: ~4 090 columns
    at ge = 2), with tails and a high-scoring homolog of the longer query at the far end of such a sequence.  Pair item and tail run
    at ONE geometry in one wave and take their cell -- column frames or plain -- from the same test, so what the pair hands over is
    in the representation the tail goes on in, on either side of the limit."""
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "2")
    if lg >= 0:
        monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    qs = synth.make_queries([180, 260], seed=901)
    rng = np.random.default_rng(902)
    seqs = [synth.random_residues(9000 + i, 0, int(l)) for i, l in enumerate(rng.integers(4060, 4100, size=40))]
    seqs += [synth.random_residues(9100 + i, 0, int(l)) for i, l in enumerate(rng.integers(3900, 4060, size=20))]
    for k in (3, 17, 44):
        mut = synth.mutate(qs[1], 0.03, 900 + k)                                    # near-copies of the LONGER query, at the end of the block
        seqs[k][-len(mut):] = mut
    seqs[29][:len(qs[0])] = qs[0]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want[1].max() > 1000


def test_query_pairs_overflow_rerun(hip_ctx, oracle, monkeypatch):
    """Either query of a pair can hit the int16 ceiling on its own."""
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    w = synth.ALPHABET.index("W")
    qa, qb = np.full(3100, w, np.uint8), np.full(2900, w, np.uint8)
    seqs = [np.full(k, w, np.uint8) for k in (5, 2978, 2979, 3100)] + [synth.random_residues(3, 0, 300)]
    from helpers import db_from_sequences
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, [qa, qb], b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, [qa, qb], b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want[0].max() == 34100 and want[1].max() == 31900


def test_pairing_is_invisible(hip_ctx, oracle, monkeypatch):
    """Default pairing rule vs. no pairing: identical score tables."""
    qs = synth.make_queries([90, 100, 310, 330, 500], seed=77)
    L, R, O = random_db(400, seed=78, max_len=300, queries=qs, homologs=1)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("pam70")
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "0")
    off = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 12, 3)
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "1")
    on = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 12, 3)
    np.testing.assert_array_equal(on, off)
    np.testing.assert_array_equal(on, expect(oracle, qs, b, n, disp, 16, sm, 12, 3))


@pytest.mark.parametrize("go,ge", [(10, 2), (0, 0), (5, 0), (3, 1)])
@pytest.mark.parametrize("seq_len,qlen", [(9000, 300), (30000, 150), (65520, 120)])   # 65 520: the longest sequence the formats allow
def test_very_long_sequences_with_multi_round_queries(hip_ctx, oracle, seq_len, qlen, go, ge):
    """Blocks longer than a wave's spill region holds at G = 1 (4096 columns): the planner must pick a
    geometry with fewer lanes per group, and the boundary row still has to come back exactly.
    Gap extend 0 and 1 keep the frame offset (columns x ge) small however long the block is, so these are
    also the blocks that would overrun the column-frame cell's floor table (OSW_I16S_TABLE columns) if the
    kernel did not send them to the plain biased cell: homologs sit past column 8 500."""
    q = synth.random_residues(5, 0, qlen)
    long_seq = synth.random_residues(6, 0, seq_len)
    hom = synth.mutate(q, 0.1, 3)[:qlen]
    long_seq[seq_len // 2:seq_len // 2 + len(hom)] = hom
    long_seq[8600:8600 + len(hom) - 20] = hom[20:]            # a second one just past the table's end
    other = synth.random_residues(7, 0, seq_len - 17)
    other[seq_len - 17 - len(hom):] = hom                     # and one that ends with the sequence
    seqs = [synth.random_residues(200 + i, 0, 40 + i) for i in range(20)] + [long_seq, other]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, [q, q[:77]], b, n, disp, 16, sm, go, ge)
    want = expect(oracle, [q, q[:77]], b, n, disp, 16, sm, go, ge)
    np.testing.assert_array_equal(got, want)
    assert want.max() > 300


def test_two_devices_in_one_context(oracle, two_devices):
    """The reference drives several accelerators from one thread: chunk c of a round goes to device
    c mod ndev, all are awaited together (FPGAsearch.c:132-138, :223).  Two context devices -- mapped onto
    the one GPU of the test box, and GPUs 0 and 1 where the box has two -- exercise that path: per-device
    streams, buffers, queues and downloads."""
    from oswald_amd import capi
    qs = synth.make_queries([150, 61, 300, 290], seed=41)
    L, R, O = random_db(900, seed=43, max_len=260, queries=qs, homologs=2)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    sm = submat.load("blosum62")
    plan = dblayout.chunk_plan(nfull, 16, 40000, 2)
    assert len(plan) >= 3
    a, m, ad = pack_queries(qs)
    with capi.Context(2, two_devices) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        parts = [None] * len(plan)
        for k in range(0, len(plan), 2):
            live = []
            for d in range(min(2, len(plan) - k)):
                g0, g1 = plan[k + d]
                b, n, disp = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
                out = np.full((len(qs), len(n) * 16), -3, np.int32)
                ctx.search_chunk_async(b, n, disp.astype(np.uint32), out, 16, dev=d)
                live.append((k + d, out, b, n, disp))  # keep the inputs alive until wait()
            ctx.wait()
            for idx, out, *_ in live:
                parts[idx] = out
    want = expect(oracle, qs, bfull, nfull, dfull.astype(np.uint32), 16, sm, 10, 2)
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), want)


def test_async_uploads_and_reserved_work_space(oracle, two_devices):
    """oswald_hip_chunk_upload_async on two context devices (uploads queued on both before either search), work
    space reserved up front for the longest sequence and grown when a later chunk holds a longer one."""
    from oswald_amd import capi
    qs = synth.make_queries([700, 64, 1500], seed=91)
    L, R, O = random_db(600, seed=92, max_len=500, queries=qs[:2], homologs=2)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    plan = dblayout.chunk_plan(nfull, 16, 60000, 2)
    assert len(plan) >= 4
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = expect(oracle, qs, bfull, nfull, dfull.astype(np.uint32), 16, sm, 10, 2)
    with capi.Context(2, two_devices) as ctx:
        ctx.reserve(500)
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        parts = [None] * len(plan)
        for k in range(0, len(plan), 2):
            live = []
            for d in range(min(2, len(plan) - k)):
                g0, g1 = plan[k + d]
                b, n, disp = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
                live.append((k + d, d, ctx.chunk_upload(b, n, disp.astype(np.uint32), 16, dev=d, wait=False), np.full((len(qs), len(n) * 16), -3, np.int32)))
            for idx, d, h, out in live:
                ctx.chunk_search(h, out, dev=d)
            ctx.wait()
            for idx, d, h, out in live:
                parts[idx] = out
                ctx.chunk_release(h, dev=d)
        np.testing.assert_array_equal(np.concatenate(parts, axis=1), want)
        # a chunk with a much longer sequence than reserved for: the work space grows (multi-round query: it is used)
        long_seq = synth.random_residues(93, 0, 5000)
        long_seq[2500:2500 + 700] = qs[0]
        L2, R2, O2 = db_from_sequences([long_seq, synth.random_residues(94, 0, 4000)] + [synth.random_residues(95 + i, 0, 90) for i in range(20)])
        b2, n2, d2, _, _ = layout(L2, R2, O2, 16)
        out2 = np.zeros((len(qs), len(n2) * 16), np.int32)
        ctx.search_chunk_async(b2, n2, d2, out2, 16, dev=0)
        ctx.wait()
        np.testing.assert_array_equal(out2, expect(oracle, qs, b2, n2, d2, 16, sm, 10, 2))


def test_scores_around_the_int16_ceiling(hip_ctx, oracle):
    """The packed-int16 cell is exact below 30576 (its values carry a bias of 1024 and must stay below the fp16
    inf/NaN patterns, see ArithI16B); sequences at or above it are re-run in int32.  Self-alignments scoring just
    below, at and above that threshold and up to the int16 limit must all come back exact."""
    targets = [30500, 30574, 30575, 30576, 30577, 30640, 31743, 32000, 32767]
    qs = [_self_scoring(t, 300 + t) for t in targets[::2]]          # 5 long queries ...
    seqs = [_self_scoring(t, 300 + t) for t in targets]             # ... against all 9 sequences (5 are exact copies)
    seqs += [synth.random_residues(1900 + i, 0, 400 + 31 * i) for i in range(12)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want.max(axis=1).tolist() == targets[::2]
    assert ((want > 30000) & (want < 30576)).any() and (want >= 30576).any()


@pytest.mark.parametrize("ge,go", [(2, 10), (0, 0), (1, 11), (7, 3)])
def test_scores_around_the_frame_cell_ceiling(hip_ctx, oracle, ge, go):
    """The column-frame int16 cell (default) hands sequences scoring 22256 or more to the int32 kernel; blocks whose
    frame offset (columns + 2G + 2) * ge would pass 8192 run on the plain biased cell (30576).  Self-alignments just
    below, at and above both thresholds, with several gap-extend penalties (ge = 7 makes the long blocks ineligible)."""
    targets = [22200, 22255, 22256, 22257, 22300, 30575, 30576, 31000]
    qs = [_self_scoring(t, 500 + t) for t in targets[1::3]]         # 22255, 22300, 31000
    seqs = [_self_scoring(t, 500 + t) for t in targets]
    seqs += [synth.random_residues(2900 + i, 0, 300 + 29 * i) for i in range(12)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge)
    want = expect(oracle, qs, b, n, disp, 16, sm, go, ge)
    np.testing.assert_array_equal(got, want)
    assert want.max(axis=1).tolist() == targets[1::3]


@pytest.mark.parametrize("matrix,go,ge", [("blosum62", 10, 2), ("pam250", 14, 2), ("blosum45", 0, 0), ("pam30", 30, 5)])
@pytest.mark.parametrize("nq", [1, 2, 5])
def test_int8_first_pass_with_int16_rerun(hip_ctx, oracle, matrix, go, ge, nq):
    """cell_bits = 8 (BASELINE configs[2]): the SWAR 8-bit first pass on query pairs, everything that leaves its
    7-bit range re-run by the packed-int16 kernel (and beyond that in int32); an odd query runs in int16.  Scores
    must be the exact ones for every (query, sequence), whatever tier computed them."""
    qs = synth.make_queries([50, 129, 375, 31, 700][:nq], seed=5)
    L, R, O = random_db(300, seed=77, max_len=400, queries=qs, homologs=3)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load(matrix)
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, go, ge, cell_bits=8, resident=True)
    want = expect(oracle, qs, b, n, disp, 16, sm, go, ge)
    np.testing.assert_array_equal(got, want)
    to16, to32 = hip_ctx.rerun_counts()
    assert want.max() > 127 and (to16 >= 1 or nq == 1)   # the planted copies are beyond the 8-bit range (one query alone: no pair, int16 throughout)


@pytest.mark.parametrize("lg", [0, 2, 4, 6])
def test_int8_every_geometry_and_thresholds(hip_ctx, oracle, lg, monkeypatch):
    """Forced wave geometries, multi-round queries (spill of the 8-bit boundary row) and self-alignments scoring just
    below, at and above the point where the 8-bit range ends (127 - bias and 127)."""
    monkeypatch.setenv("OSWALD_HIP_FORCE_LG", str(lg))
    targets = [110, 118, 122, 123, 124, 126, 127, 128, 140]
    qs = [_self_scoring(t, 900 + t) for t in targets[::2]] + [synth.random_residues(4, 0, 300)]
    seqs = [_self_scoring(t, 900 + t) for t in targets] + [synth.random_residues(3000 + i, 0, 30 + 11 * i) for i in range(40)]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2, cell_bits=8)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want.max(axis=1)[:5].tolist() == targets[::2]


def test_int8_falls_back_to_int16_when_penalties_do_not_fit(hip_ctx, oracle):
    """Gap penalties above 127 do not fit the 8-bit cells (the reference's int8 kernels wrap there,
    HybridSearch.c:1520): the search runs on the int16 cells and stays exact."""
    qs = synth.make_queries([60, 90], seed=3)
    L, R, O = random_db(150, seed=4, max_len=200, queries=qs, homologs=2)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 200, 70, cell_bits=8)
    np.testing.assert_array_equal(got, expect(oracle, qs, b, n, disp, 16, sm, 200, 70))
    assert hip_ctx.rerun_counts()[0] == 0


def test_api_misuse_of_the_round2_entry_points(hip_ctx):
    """Errors are reported through the return code and oswald_hip_last_error(), never swallowed."""
    from oswald_amd import capi
    sm = submat.load("blosum62")
    qs = synth.make_queries([40], seed=2)
    a, m, ad = pack_queries(qs)
    L, R, O = random_db(50, seed=3, max_len=60)
    b, n, disp, _, _ = layout(L, R, O, 16)
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.reserve(1000, dev=7)                       # device index out of range
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.set_scoring(sm, 10, 2, 12)                  # cell_bits must be 0, 8, 16 or 32
    h = hip_ctx.chunk_upload(b, n, disp, 16, wait=False)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_topr(h, 50, 3)                        # not searched yet
    hip_ctx.chunk_release(h)                                # releasing a chunk whose upload is still queued is fine
    hip_ctx.wait()
    hip_ctx.reserve(65520)                                  # the longest sequence the formats allow
    out = np.zeros((1, len(n) * 16), np.int32)
    hip_ctx.search_chunk_async(b, n, disp, out, 16)
    hip_ctx.wait()
    assert out.max() > 0


@pytest.mark.parametrize("bits", [0, 8])
def test_longest_queries(hip_ctx, oracle, bits):
    """Queries of 30 000 and 65 535 residues (the longest a uint16 length allows; the reference's FPGA kernel stops at
    5 478, sw.cl:5): hundreds of strips per lane group, many rounds, and a second query far shorter (pairs of very
    different lengths in the 8-bit mode)."""
    qs = [synth.random_residues(71, 0, 65535), synth.random_residues(72, 0, 30000), synth.random_residues(73, 0, 12)]
    seqs = [synth.random_residues(300 + i, 0, 30 + 13 * i) for i in range(24)]
    seqs.append(qs[1][10000:10400].copy())                # a 400-residue piece of the second query: exact score 2000+
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2, cell_bits=bits)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want[1].max() > 2000


@pytest.mark.parametrize("bits", [0, 8])
def test_many_queries_in_one_set(hip_ctx, oracle, bits):
    """150 queries in one set (the reference stops at 100, utils.c:135): 75 pairs or a mix of pairs and single
    queries, thousands of work items in both queues."""
    rng = np.random.default_rng(44)
    qs = [synth.random_residues(5000 + i, 0, int(l)) for i, l in enumerate(rng.integers(20, 260, 150))]
    L, R, O = random_db(400, seed=45, max_len=220, queries=qs[:6], homologs=1)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2, cell_bits=bits)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dealt", [False, True])
def test_context_level_topr_over_chunks_and_devices(oracle, dealt, two_devices):
    """oswald_hip_topr (reference FPGAsearch.c:236-237 merge, :312-321 + utils.c:71-86 sort): two context
    devices, five chunks in three rounds (slots re-used: the lists are collected at search time), r larger than some
    chunks; the merged list equals the top r of the whole score table in the reference's order -- with chunks that are
    contiguous runs (first_index) and with chunks dealt block-wise (index map).  Ties are plentiful (short sequences).
    The lists are folded on the devices; with two distinct GPUs they cross over RCCL (comm_info says how many ranks the
    communicator saw).  The host-side merge of the library is only the checker here."""
    from oswald_amd import capi, multigpu
    qs = synth.make_queries([33, 60, 61, 150], seed=141)
    NSEQ = 1400
    L, R, O = random_db(NSEQ, seed=143, max_len=70, queries=qs[-1:], homologs=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    whole = expect(oracle, qs, bfull, nfull, dfull.astype(np.uint32), 16, sm, 10, 2)
    if dealt:   # five "ranks'" shares of the database as five chunks: none is a contiguous run
        pieces = [multigpu.dealt_positions(NSEQ, 5, k) for k in range(5)]
    else:
        plan = dblayout.chunk_plan(nfull, 16, 16 * int(nfull.sum()) // 5 + 1, 1)
        assert len(plan) >= 5
        pieces = [np.arange(g0 * 16, min(g1 * 16, NSEQ)) for g0, g1 in plan]
    r = 300
    with capi.Context(2, two_devices) as ctx:
        assert ctx.comm_info()["context_ranks"] == len(set(two_devices))
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        for rep in range(2):   # a second collection starts from scratch
            ctx.topr_begin(r)
            for k in range(0, len(pieces), 2):
                live = []
                for d in range(min(2, len(pieces) - k)):
                    pos = pieces[k + d]
                    ls = sl[pos]
                    off = np.zeros(len(pos) + 1, np.int64)
                    np.cumsum(ls, out=off[1:])
                    res = np.concatenate([sr[so[p]:so[p + 1]] for p in pos])
                    b, n, disp = dblayout.interleave(ls, res, off, 16)
                    h = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16, dev=d, wait=False)
                    if dealt:
                        ctx.chunk_set_index(h, 0, len(pos), pos, dev=d)
                    else:
                        ctx.chunk_set_index(h, int(pos[0]), len(pos), None, dev=d)
                    live.append((d, h, b, n, disp))
                for d, h, *_ in live:
                    ctx.chunk_search(h, None, dev=d)
                for d, h, *_ in live:
                    ctx.chunk_release(h, dev=d)
            for rr in (r, 10):
                sc, ix = ctx.topr(rr)
                for q in range(len(qs)):
                    ws, wi = dblayout.topr_reference_order(whole[q, :NSEQ], rr)
                    np.testing.assert_array_equal(sc[q], ws)
                    np.testing.assert_array_equal(ix[q], wi)
        with pytest.raises(capi.OswaldHipError):
            ctx.topr(r + 1)            # more than was collected
        with pytest.raises(capi.OswaldHipError):
            ctx.topr_begin(5000)       # the device selection stops at 1024
    assert len(np.unique(whole[0, :NSEQ])) < NSEQ // 4   # ties are present


@pytest.mark.parametrize("seq_len,qlen,ge", [(30000, 150, 2), (65520, 120, 1)])
def test_very_long_sequences_on_the_int32_cells(hip_ctx, oracle, seq_len, qlen, ge):
    """cell_bits = 32 on blocks far longer than the int16 frame cell's floor table (8 448 columns): the hand-scheduled int32 cell
    has a floor table of its own for every column the formats allow (OSW_I32F_TABLE), read by the first round of every item;
    the planner must still pick a geometry whose boundary row fits the spill region."""
    q = synth.random_residues(15, 0, qlen)
    long_seq = synth.random_residues(16, 0, seq_len)
    hom = synth.mutate(q, 0.1, 3)[:qlen]
    long_seq[seq_len // 2:seq_len // 2 + len(hom)] = hom
    long_seq[8600:8600 + len(hom) - 20] = hom[20:]
    other = synth.random_residues(17, 0, seq_len - 17)
    other[seq_len - 17 - len(hom):] = hom                     # one that ends with the sequence
    seqs = [synth.random_residues(300 + i, 0, 40 + i) for i in range(20)] + [long_seq, other]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    qs = [q, q[:77], synth.random_residues(18, 0, 700)]       # (700 rows: several rounds at any geometry)
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, ge, cell_bits=32)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, ge)
    np.testing.assert_array_equal(got, want)
    assert want.max() > 300


def test_int32_rerun_of_a_copy_inside_a_very_long_sequence(hip_ctx, oracle):
    """A re-run item whose sequence is far longer than the query: a near-copy of a 6 600-row query in the middle of a 22 600-column
    sequence, and one at the very end of another.  Blocks that long run on the plain biased int16 cell (their frame offsets would
    not fit the column-frame cell), whose ceiling is 30 576: the copies score beyond it.  The pipeline's waves then hand
    22 000-column boundary rows to each other, the compact copy of the residues (behind the columns of the workgroup's first spill
    region) is as long as it gets in practice, and the first round reads the int32 floor table far beyond the int16 one's 8 448
    entries."""
    q = synth.make_queries([6600], seed=311)[0]
    rng = np.random.default_rng(313)
    mid = np.concatenate([synth.random_residues(41, 0, 8000), synth.mutate(np.asarray(q, np.uint8), 0.01, 5), synth.random_residues(42, 0, 8000)])
    end = np.concatenate([synth.random_residues(43, 0, 15000), synth.mutate(np.asarray(q, np.uint8), 0.02, 6)])
    seqs = [synth.random_residues(9100 + i, 0, int(l)) for i, l in enumerate(rng.integers(30, 900, size=40))] + [mid, end]
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 16)
    sm = submat.load("blosum62")
    qs = [q, q[:333]]
    got = run_gpu(hip_ctx, qs, b, n, disp, 16, sm, 10, 2)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    assert want.max() > 30576 and hip_ctx.rerun_counts()[1] >= 1   # (the two sequences are a lane pair: one item, both halves)


@pytest.mark.parametrize("qlens", [[4300], [6100, 1500], [2990, 5000, 3700]])
def test_int32_rerun_pipeline_over_many_long_items(hip_ctx, oracle, qlens):
    """The int32 re-run as a workgroup pipeline (osw_sw_i32, round 4): wave w of a workgroup runs rounds w, w+4, ... of ONE
    (query, sequence), the boundary row going from wave to wave through the spill scratch.  Near-copies of long queries
    (self-scores 16 000 .. 33 000: beyond the column-frame cell's 22 256, some beyond the plain cell's 30 576), next to
    each other in the sorted database so that both sequences of a lane pair and several lanes of a block are re-run,
    queries of 12 .. 24 rounds (not multiples of four), shorter sequences around them; every score against the AVX2 port
    (int8 -> int16 -> int32 like the reference's host path)."""
    qs = synth.make_queries(qlens, seed=171)
    rng = np.random.default_rng(173)
    seqs = [synth.random_residues(9000 + i, 0, int(l)) for i, l in enumerate(rng.integers(30, 900, size=150))]
    k = 0
    for q in qs:
        for rate in (0.0, 0.02, 0.05, 0.10, 0.20, 0.35):
            for cut in (0, 137):                      # whole copies and copies that lack their first residues
                seqs.append(synth.mutate(np.asarray(q[cut:], np.uint8), rate, 4242 + k))
                k += 1
    L, R, O = db_from_sequences(seqs)
    b, n, disp, _, _ = layout(L, R, O, 32, round_to=1)
    sm = submat.load("blosum62")
    got = run_gpu(hip_ctx, qs, b, n, disp, 32, sm, 10, 2, resident=True)
    a, m, ad = pack_queries(qs)
    want, stage = oracle.search_chunk_simd(a, m, ad, b, n, disp, 32, sm, 10, 2)
    np.testing.assert_array_equal(got, want)
    if max(qlens) >= 6000:      # beyond the ceiling of either int16 cell: the int32 tier ran
        assert want.max() > 30576 and hip_ctx.rerun_counts()[1] >= 1


def test_process_level_communicator_at_world_size_one(oracle):
    """oswald_hip_comm_unique_id / _comm_init_rank / _comm_info: the gather of a multi-process job (one rank per GPU;
    SURVEY 8e) runs inside the C ABI -- ncclCommInitRank, then every oswald_hip_topr all-gathers the ranks' lists and
    folds them on the GPU.  One GPU in the box = a world of one rank: the RCCL calls are the real ones, the result must
    be the list the context finds by itself, and RCCL itself must report one rank."""
    from oswald_amd import capi
    qs = synth.make_queries([33, 60, 150], seed=151)
    NSEQ = 700
    L, R, O = random_db(NSEQ, seed=153, max_len=90, queries=qs[-1:], homologs=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    whole = expect(oracle, qs, b, n, disp.astype(np.uint32), 16, sm, 10, 2)
    with capi.Context(1, [0]) as ctx:
        info = ctx.comm_info()
        assert info["context_ranks"] == 1 and info["process_ranks"] == 0 and info["process_rank"] == -1 and info["rccl_version"] > 20000
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        h = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
        ctx.chunk_set_index(h, 0, NSEQ)
        ctx.topr_begin(25)
        ctx.chunk_search(h, None)
        alone = ctx.topr(25)
        ident = capi.comm_unique_id()
        assert len(ident) == capi.COMM_ID_BYTES and any(ident)
        with pytest.raises(capi.OswaldHipError):
            ctx.comm_init_rank(ident, 1, 1)          # rank out of range
        ctx.comm_init_rank(ident, 1, 0)
        info = ctx.comm_info()
        assert info["process_ranks"] == 1 and info["process_rank"] == 0
        with pytest.raises(capi.OswaldHipError):
            ctx.comm_init_rank(ident, 1, 0)          # a context joins one job
        for rep in range(2):
            ctx.topr_begin(25)
            ctx.chunk_search(h, None)
            sc, ix = ctx.topr(25)
            np.testing.assert_array_equal(sc, alone[0])
            np.testing.assert_array_equal(ix, alone[1])
        for q in range(len(qs)):
            ws, wi = dblayout.topr_reference_order(whole[q, :NSEQ], 25)
            np.testing.assert_array_equal(sc[q], ws)
            np.testing.assert_array_equal(ix[q], wi)


def test_process_level_communicator_two_ranks_on_two_gpus(oracle):
    """Two ranks of a job, one GPU each, joined by oswald_hip_comm_init_rank (here: two threads of this process, each
    with a context of its own, where the box has two GPUs; skipped on the one-GPU test box, runs by itself on a
    multi-GPU node).  Each rank searches half of the database under the dealt shard rule; oswald_hip_topr all-gathers the
    two lists over RCCL and BOTH ranks must come back with the top r of the whole database in the reference's order."""
    import threading
    from oswald_amd import capi, multigpu
    if capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % capi.device_count())
    qs = synth.make_queries([33, 60, 150], seed=161)
    NSEQ = 1500
    L, R, O = random_db(NSEQ, seed=163, max_len=80, queries=qs[-1:], homologs=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    whole = expect(oracle, qs, bfull, nfull, dfull.astype(np.uint32), 16, sm, 10, 2)
    ident = capi.comm_unique_id()
    r, out, err = 40, [None, None], [None, None]

    def rank(k):
        try:
            pos = multigpu.dealt_positions(NSEQ, 2, k)
            ls = sl[pos]
            off = np.zeros(len(pos) + 1, np.int64)
            np.cumsum(ls, out=off[1:])
            res = np.concatenate([sr[so[p]:so[p + 1]] for p in pos])
            b, n, disp = dblayout.interleave(ls, res, off, 16)
            with capi.Context(1, [k]) as ctx:
                ctx.comm_init_rank(ident, 2, k)
                info = ctx.comm_info()
                assert info["process_ranks"] == 2 and info["process_rank"] == k
                ctx.set_scoring(sm, 10, 2)
                ctx.set_queries(a, m, ad)
                h = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
                ctx.chunk_set_index(h, 0, len(pos), pos)
                for rep in range(2):
                    ctx.topr_begin(r)
                    ctx.chunk_search(h, None)
                    out[k] = ctx.topr(r)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            err[k] = e

    th = [threading.Thread(target=rank, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a rank hangs in the collective"
    assert err == [None, None], err
    for k in range(2):
        sc, ix = out[k]
        for q in range(len(qs)):
            ws, wi = dblayout.topr_reference_order(whole[q, :NSEQ], r)
            np.testing.assert_array_equal(sc[q], ws)
            np.testing.assert_array_equal(ix[q], wi)


def test_api_misuse_of_the_round3_entry_points(hip_ctx):
    """oswald_hip_topr / _topr_begin / _chunk_set_index / _merge_candidates report misuse through the return code."""
    from oswald_amd import capi
    sm = submat.load("blosum62")
    qs = synth.make_queries([40, 41], seed=2)
    a, m, ad = pack_queries(qs)
    L, R, O = random_db(50, seed=3, max_len=60)
    b, n, disp, _, _ = layout(L, R, O, 16)
    hip_ctx.set_scoring(sm, 10, 2)
    hip_ctx.set_queries(a, m, ad)
    hip_ctx.topr_begin(0)                                       # collection switched off
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.topr(5)                                         # nothing was being collected
    h = hip_ctx.chunk_upload(b, n, disp, 16)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_set_index(h, 0, len(n) * 16 + 1)          # more sequences than the chunk has lanes
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_set_index(h + 7, 0, 50)                   # no such chunk
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_set_index(h, 0xFFFFFFF0, 50)              # database indices must fit 32 bits
    hip_ctx.topr_begin(5)
    hip_ctx.chunk_search(h, None)                               # no index given: the chunk is searched but not collected
    sc, ix = hip_ctx.topr(5)
    assert (sc == -1).all() and (ix == 0xFFFFFFFF).all()
    hip_ctx.chunk_set_index(h, 1000, 50)
    hip_ctx.chunk_search(h, None)
    once = hip_ctx.topr(4)
    hip_ctx.chunk_search(h, None)                               # searched twice: counts once (lists hold distinct database keys)
    sc, ix = hip_ctx.topr(4)
    assert (ix[:, 0] != ix[:, 1]).all() and (ix >= 1000).all() and (ix < 1050).all()
    np.testing.assert_array_equal(sc, once[0])
    np.testing.assert_array_equal(ix, once[1])
    hip_ctx.set_queries(a[:40], m[:1], ad[:1])                  # the query set changes under a collection
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.topr(4)
    with pytest.raises(capi.OswaldHipError):
        hip_ctx.chunk_search(h, None)                           # ... and a search would fold lists of another query set in
    hip_ctx.topr_begin(4)                                       # a new collection for the new set
    hip_ctx.chunk_search(h, None)
    s1, i1 = hip_ctx.topr(4)
    assert s1.shape == (1, 4) and (s1[0] == once[0][0]).all()
    hip_ctx.topr_begin(0)
    hip_ctx.chunk_release(h)
    s2, i2 = capi.merge_candidates(np.array([[5, -1, 5, 7]], np.int32), np.array([[3, 0, 9, 1]], np.uint32), 3)
    assert s2.tolist() == [[7, 5, 5]] and i2.tolist() == [[1, 9, 3]]


@pytest.mark.parametrize("W", [16, 64])
@pytest.mark.parametrize("bits", [16, 8, 32])
def test_residue_codes_beyond_the_alphabet_score_like_the_dummy(hip_ctx, oracle, W, bits):
    """The reference's preprocessing emits the codes 0..23 and its matrices hold zeros in column 23 (the dummy) and in the
    padding columns 24..31 (host/src/submat.c).  The re-tile kernels store a code >= 24 as 23 -- the single-query kernels'
    profile has 24 entries per row-block -- which is exact for such matrices: one query, a pair and an odd one out against
    sequences sprinkled with the codes 24..31, on every cell mode and both re-tile kernels (W = 16 has its own)."""
    rng = np.random.default_rng(2404)
    for qlens in ([120], [60, 61], [33, 90, 91]):
        qs = synth.make_queries(qlens, seed=3)
        seqs = [rng.integers(0, 24, int(l)).astype(np.uint8) for l in rng.integers(1, 300, 150)]
        for s in seqs[::3]:
            s[rng.integers(0, len(s), max(1, len(s) // 5))] = rng.integers(24, 32, max(1, len(s) // 5))
        seqs.append(np.concatenate([qs[-1], rng.integers(24, 32, 7).astype(np.uint8), qs[-1]]))
        L, R, O = db_from_sequences(seqs)
        b, n, disp, _, _ = layout(L, R, O, W)
        assert (b >= 24).any()
        sm = submat.load("blosum62")
        got = run_gpu(hip_ctx, qs, b, n, disp, W, sm, 10, 2, cell_bits=bits)
        want = expect(oracle, qs, b, n, disp, W, sm, 10, 2)
        np.testing.assert_array_equal(got, want)


def test_matrix_padding_columns_must_repeat_the_dummy_column(hip_ctx):
    """... and a matrix for which that would not be exact is refused by oswald_hip_set_scoring, with the reason."""
    from oswald_amd import capi
    sm = submat.load("blosum62").copy()
    sm[5, 27] = 3
    with pytest.raises(capi.OswaldHipError, match="column 27"):
        hip_ctx.set_scoring(sm, 10, 2, 0)
    sm = submat.load("blosum62").copy()
    sm[:, 23:] = -2   # any value, as long as the padding repeats column 23
    hip_ctx.set_scoring(sm, 10, 2, 0)
