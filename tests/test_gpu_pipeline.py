"""The upload pipeline of the C ABI (GPU): searches planned and queued while their chunk's upload is still on its way
(work queues planned on the group-length extents, launches ordered behind the upload on the device), slots re-used
beside a running search (two sets of work queues per slot), score tables and top lists against the oracle.

Reference: the per-chunk loop of fpga_search (uploads, launches, download: host/src/FPGAsearch.c:180-238)."""
import numpy as np
import pytest

from oswald_amd import dblayout, submat, synth
from helpers import pack_queries, random_db
from test_gpu_parity import expect

pytestmark = pytest.mark.gpu


_WANT = {}   # the oracle's table of a fixture, computed once for all the modes


def _want(key, oracle, qs, bfull, nfull, dfull, sm, go, ge):
    if key not in _WANT:
        _WANT[key] = expect(oracle, qs, bfull, nfull, dfull, 16, sm, go, ge)
    return _WANT[key]


def _chunks(nseq, qs, max_chunk, seed):
    L, R, O = random_db(nseq, seed=seed, min_len=40, max_len=420, queries=qs, homologs=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    bfull, nfull, dfull = dblayout.interleave(sl, sr, so, 16)
    plan = dblayout.chunk_plan(nfull, 16, max_chunk, 1)
    parts = []
    for g0, g1 in plan:
        b, n, disp = dblayout.interleave(sl, sr, so, 16, g_begin=g0, g_end=g1)
        parts.append((b, n, disp.astype(np.uint32), g0 * 16, min(len(sl), g1 * 16) - g0 * 16))
    return bfull, nfull, dfull.astype(np.uint32), parts


@pytest.mark.parametrize("mode", ["in flight", "group-length extents always", "plan waits for the upload"])
@pytest.mark.parametrize("ahead", [1, 2])
def test_searches_queued_behind_uploads_in_flight(oracle, monkeypatch, mode, ahead):
    """Every chunk is uploaded with _async `ahead` chunks before it is searched, searched without waiting for the upload,
    and released right behind the search, so that the next upload re-uses its slot while its search may still be running.
    Score tables and the context-level top lists must be those of the whole database searched in one piece, whether the
    work queues were planned on the live extents or on the group lengths."""
    from oswald_amd import capi
    if mode == "group-length extents always":
        monkeypatch.setenv("OSWALD_HIP_PLAN_EST", "1")
    elif mode == "plan waits for the upload":
        monkeypatch.setenv("OSWALD_HIP_PLAN_WAITS", "1")
    qs = synth.make_queries([96, 171, 260, 333, 410, 518, 77], seed=611)
    bfull, nfull, dfull, parts = _chunks(3000, qs, 120000, seed=612)
    assert len(parts) >= 6
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("pipeline", oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    nvalid = sum(p[4] for p in parts)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        for rep in range(3):  # (the later passes find every slot and both sets of work queues in place)
            outs = [np.full((len(qs), len(p[1]) * 16), -3, np.int32) for p in parts]
            ctx.topr_begin(10)
            hs = {}
            for k in range(min(ahead, len(parts))):
                hs[k] = ctx.chunk_upload(parts[k][0], parts[k][1], parts[k][2], 16, wait=False)
                ctx.chunk_set_index(hs[k], parts[k][3], parts[k][4])
            for k in range(len(parts)):
                ctx.chunk_search(hs[k], outs[k])
                ctx.chunk_release(hs[k])
                j = k + ahead
                if j < len(parts):
                    hs[j] = ctx.chunk_upload(parts[j][0], parts[j][1], parts[j][2], 16, wait=False)
                    ctx.chunk_set_index(hs[j], parts[j][3], parts[j][4])
            sc, ix = ctx.topr(10)
            ctx.wait()
            got = np.concatenate(outs, axis=1)
            np.testing.assert_array_equal(got, want)
            # top lists: descending score, ties by descending database index (utils.c:3-86), over the real sequences
            for q in range(len(qs)):
                row = want[q, :nvalid].astype(np.int64)
                key = np.sort((row << 32) | np.arange(nvalid))[::-1][:10]
                np.testing.assert_array_equal(sc[q], (key >> 32).astype(np.int32))
                np.testing.assert_array_equal(ix[q].astype(np.int64), key & 0xFFFFFFFF)


def test_resident_chunk_is_replanned_on_its_live_extents(oracle):
    """A chunk whose first search was planned on the group lengths (upload in flight) is planned again on the live extents
    at its next search; both searches give the same table."""
    from oswald_amd import capi
    qs = synth.make_queries([200, 345], seed=71)
    bfull, nfull, dfull, parts = _chunks(2500, qs, 1 << 30, seed=72)
    assert len(parts) == 1
    sm = submat.load("pam250")
    a, m, ad = pack_queries(qs)
    want = expect(oracle, qs, bfull, nfull, dfull, 16, sm, 14, 2)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 14, 2)
        ctx.set_queries(a, m, ad)
        h = ctx.chunk_upload(parts[0][0], parts[0][1], parts[0][2], 16, wait=False)
        for _ in range(3):
            out = np.full((len(qs), len(nfull) * 16), -3, np.int32)
            ctx.chunk_search(h, out)
            ctx.wait()
            np.testing.assert_array_equal(out, want)


def test_pipeline_beside_running_searches(oracle):
    """The same flow with searches long enough (milliseconds each) for the next chunk's plan and upload to arrive while a
    search is running: 20 queries x 60 000 sequences in chunks of 6 MiB, uploads two chunks ahead.  Checked against the
    oracle's SIMD port (itself pinned to the reference's goldens in tests/test_oracle_golden.py)."""
    from oswald_amd import capi, multigpu
    qlens = synth.default_query_lengths()
    qs = synth.make_queries(qlens)
    plan = synth.DatabasePlan(60000, qs, synth.SEED_DB, 12)
    shard = multigpu.ShardedDatabase(plan, 16, 6 << 20, 1, 0, "reference")
    chunks = [shard.chunk(k) for k in range(len(shard.mine))]
    assert len(chunks) >= 4
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = [oracle.search_chunk_simd(a, m, ad, c["b"], c["n"], c["disp"].astype(np.uint32), 16, sm, 10, 2, 256, oracle.max_threads())[0] for c in chunks]
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        for rep in range(3):
            outs = [np.full((len(qs), len(c["n"]) * 16), -3, np.int32) for c in chunks]
            hs = {k: ctx.chunk_upload(chunks[k]["b"], chunks[k]["n"], chunks[k]["disp"], 16, wait=False) for k in range(2)}
            for k in range(len(chunks)):
                ctx.chunk_search(hs[k], outs[k])
                ctx.chunk_release(hs[k])
                if k + 2 < len(chunks):
                    hs[k + 2] = ctx.chunk_upload(chunks[k + 2]["b"], chunks[k + 2]["n"], chunks[k + 2]["disp"], 16, wait=False)
            ctx.wait()
            for k in range(len(chunks)):
                np.testing.assert_array_equal(outs[k][:, :want[k].shape[1]], want[k][:, :outs[k].shape[1]])
