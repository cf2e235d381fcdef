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


@pytest.mark.parametrize("tables", ["pageable (pinned by the library for the call)", "page-locked: written by the kernels", "page-locked, by DMA"])
def test_pipeline_beside_running_searches(oracle, monkeypatch, tables):
    """The same flow with searches long enough (milliseconds each) for the next chunk's plan and upload to arrive while a
    search is running: 20 queries x 60 000 sequences in chunks of 6 MiB, uploads two chunks ahead.  Checked against the
    oracle's SIMD port (itself pinned to the reference's goldens in tests/test_oracle_golden.py).  The score tables reach the
    caller in three ways: a page-locked table (oswald_hip_host_alloc) is written by the kernels themselves as they finish their
    items (round 5), a pageable one of this size is pinned by the library for the call and written the same way, and
    OSWALD_HIP_NO_DIRECT_TABLE=1 sends tables by DMA on the download stream behind every search (the way of rounds 2 - 4, and of
    small pageable tables still)."""
    from oswald_amd import capi, multigpu
    if tables.endswith("by DMA"):
        monkeypatch.setenv("OSWALD_HIP_NO_DIRECT_TABLE", "1")
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
            if tables.startswith("pageable"):
                outs = [np.full((len(qs), len(c["n"]) * 16), -3, np.int32) for c in chunks]
            else:
                bufs = [capi.HostBuffer((len(qs), len(c["n"]) * 16), np.int32) for c in chunks]
                outs = [hb.a for hb in bufs]
                for o in outs:
                    o[...] = -3
            hs = {k: ctx.chunk_upload(chunks[k]["b"], chunks[k]["n"], chunks[k]["disp"], 16, wait=False) for k in range(2)}
            for k in range(len(chunks)):
                ctx.chunk_search(hs[k], outs[k])
                ctx.chunk_release(hs[k])
                if k + 2 < len(chunks):
                    hs[k + 2] = ctx.chunk_upload(chunks[k + 2]["b"], chunks[k + 2]["n"], chunks[k + 2]["disp"], 16, wait=False)
            ctx.wait()
            for k in range(len(chunks)):
                np.testing.assert_array_equal(outs[k][:, :want[k].shape[1]], want[k][:, :outs[k].shape[1]])


def _toplists(want, nvalid, r):
    """descending score, ties by descending database index (utils.c:3-86), over the real sequences"""
    sc, ix = [], []
    for q in range(want.shape[0]):
        key = np.sort((want[q, :nvalid].astype(np.int64) << 32) | np.arange(nvalid))[::-1][:r]
        sc.append((key >> 32).astype(np.int32))
        ix.append(key & 0xFFFFFFFF)
    return np.array(sc), np.array(ix)


@pytest.mark.parametrize("use_map", [False, True])
def test_first_chunk_cut_in_two_by_the_library(oracle, monkeypatch, use_map):
    """An asynchronous upload of some size that finds its device idle is cut by the library into a head of whole
    128-sequence blocks and the rest (two copies, two re-tiles, two searches behind one another).  Nothing of it
    shows at the boundary: one handle, one score table, one index (first index or map), chunk-level and context-level
    top lists, geometry, release -- all against the oracle, and the same chunk uploaded synchronously (never cut)."""
    from oswald_amd import capi
    monkeypatch.setenv("OSWALD_HIP_SPLIT_BYTES", "20000")
    qs = synth.make_queries([33, 150, 301], seed=81)
    bfull, nfull, dfull, parts = _chunks(2100, qs, 1 << 30, seed=82)
    assert len(parts) == 1
    b, n, disp, first, nvalid = parts[0]
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("cut", oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    rng = np.random.default_rng(5)
    index_map = rng.permutation(nvalid).astype(np.uint32) + 1000 if use_map else None
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        whole = ctx.chunk_upload(b, n, disp, 16)                    # synchronous: one piece
        g_whole = ctx.chunk_geometry(whole)
        ctx.chunk_release(whole)
        ctx.wait()
        for rep in range(3):
            ctx.topr_begin(12)
            h = ctx.chunk_upload(b, n, disp, 16, wait=False)        # the device is idle: cut in two
            ctx.chunk_set_index(h, 7, nvalid, index_map)
            out = np.full((len(qs), len(n) * 16), -3, np.int32)
            ctx.chunk_search(h, out)
            sc, ix = ctx.topr(12)
            ctx.wait()
            np.testing.assert_array_equal(out, want)
            g = ctx.chunk_geometry(h)
            assert g["blocks"] == g_whole["blocks"] and g["col4_live"] == g_whole["col4_live"]
            wsc, wix = _toplists(want, nvalid, 12)
            if use_map:   # keys are (score, database index): re-rank the oracle's table on the mapped indices
                wsc, wix = [], []
                for q in range(len(qs)):
                    key = np.sort((want[q, :nvalid].astype(np.int64) << 32) | index_map.astype(np.int64))[::-1][:12]
                    wsc.append((key >> 32).astype(np.int32)); wix.append(key & 0xFFFFFFFF)
                wsc, wix = np.array(wsc), np.array(wix)
            else:
                wix = wix + 7
            np.testing.assert_array_equal(sc, wsc)
            np.testing.assert_array_equal(ix.astype(np.int64), wix)
            csc, cix = ctx.chunk_topr(h, nvalid, 12)                # chunk-level list: indices inside the chunk
            psc, pix = _toplists(want, nvalid, 12)
            np.testing.assert_array_equal(csc, psc)
            np.testing.assert_array_equal(cix.astype(np.int64), pix)
            ctx.chunk_release(h)


def test_registered_mapping_is_uploaded_by_dma(oracle, tmp_path):
    """The CLI's path of round 5: the database lives in a read-only private file mapping (the group cache), page-locked
    in place with oswald_hip_host_register; chunks are slices of it, uploaded asynchronously two ahead, searched and
    released; tables and top lists against the oracle.  (Registered memory makes the uploads plain DMA: profiles/r05_pin_probe.txt.)"""
    from oswald_amd import capi
    qs = synth.make_queries([120, 260, 77, 410], seed=91)
    bfull, nfull, dfull, parts = _chunks(3000, qs, 150000, seed=92)
    assert len(parts) >= 5
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("registered", oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    path = tmp_path / "groups.bin"
    np.concatenate([p[0] for p in parts]).tofile(path)
    mm = np.memmap(path, dtype=np.uint8, mode="r")        # PROT_READ mapping of the file
    offs = np.concatenate([[0], np.cumsum([p[0].size for p in parts])])
    nvalid = sum(p[4] for p in parts)
    with capi.Context(1) as ctx:
        reg = capi.Registered(mm)
        try:
            ctx.set_scoring(sm, 10, 2)
            ctx.set_queries(a, m, ad)
            for rep in range(2):
                outs = [np.full((len(qs), len(p[1]) * 16), -3, np.int32) for p in parts]
                ctx.topr_begin(10)
                hs = {}
                def up(k):
                    hs[k] = ctx.chunk_upload(mm[offs[k]:offs[k + 1]], parts[k][1], parts[k][2], 16, wait=False)
                    ctx.chunk_set_index(hs[k], parts[k][3], parts[k][4])
                up(0)
                for k in range(len(parts)):
                    ctx.chunk_search(hs[k], outs[k])
                    ctx.chunk_release(hs[k])
                    for j in (k + 1, k + 2):
                        if j < len(parts) and j not in hs:
                            up(j)
                sc, ix = ctx.topr(10)
                ctx.wait()
                np.testing.assert_array_equal(np.concatenate(outs, axis=1), want)
                wsc, wix = _toplists(want, nvalid, 10)
                np.testing.assert_array_equal(sc, wsc)
                np.testing.assert_array_equal(ix.astype(np.int64), wix)
        finally:
            reg.close()


def test_replanned_chunk_does_not_overwrite_queues_in_use(oracle):
    """upload X (async); search X while the upload is in flight (plan on the group lengths, queue set 1); search X
    again at once (the upload has landed: a re-plan on the live extents would go to set 0 -- the library keeps the plan the
    running search uses); release X; upload Y into the slot; search Y -- whose plan must not land in a set a search of X is
    still pulling from.  No wait anywhere in between; many rounds; every table against the oracle."""
    from oswald_amd import capi
    qlens = synth.default_query_lengths()[:8]
    qs = synth.make_queries(qlens)
    plan = synth.DatabasePlan(24000, qs, synth.SEED_DB, 6)
    from oswald_amd import multigpu
    shard = multigpu.ShardedDatabase(plan, 16, 3 << 20, 1, 0, "reference")
    chunks = [shard.chunk(k) for k in range(len(shard.mine))]
    assert len(chunks) >= 3
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = [oracle.search_chunk_simd(a, m, ad, c["b"], c["n"], c["disp"].astype(np.uint32), 16, sm, 10, 2, 256, oracle.max_threads())[0] for c in chunks]
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        for rep in range(4):
            outs = [[np.full((len(qs), len(c["n"]) * 16), -3, np.int32) for _ in range(2)] for c in chunks]
            for k, c in enumerate(chunks):
                h = ctx.chunk_upload(c["b"], c["n"], c["disp"], 16, wait=False)
                ctx.chunk_search(h, outs[k][0])
                ctx.chunk_search(h, outs[k][1])
                ctx.chunk_release(h)
            ctx.wait()
            for k in range(len(chunks)):
                for o in outs[k]:
                    np.testing.assert_array_equal(o[:, :want[k].shape[1]], want[k][:, :o.shape[1]])


def test_one_call_search_is_asynchronous(oracle):
    """oswald_hip_search_chunk_async returns without waiting for its upload (the reference's four
    clEnqueueWriteBuffer per device are non-blocking, FPGAsearch.c:180-198): with two context devices the second call is
    made -- and returns -- while the first device's chunk is still on the link.  A chunk of ~190 MB from page-locked
    memory takes ~3.4 ms to copy; both calls together must return in a fraction of the time until the devices are through."""
    import time
    from oswald_amd import capi
    qs = synth.make_queries([24], seed=3)
    rng = np.random.default_rng(77)
    ngroups, cols = 4096, 2912                      # 4096 groups of 16 sequences of 2912 residues: 190 MB
    n = np.full(ngroups, cols, np.uint16)
    disp = (np.arange(ngroups, dtype=np.uint64) * cols * 16).astype(np.uint32)
    b = rng.integers(0, 20, size=ngroups * cols * 16, dtype=np.uint8)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = oracle.search_chunk_simd(a, m, ad, b, n, disp, 16, sm, 10, 2, 256, oracle.max_threads())[0]
    with capi.Context(2, [0, 0]) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        ctx.reserve(cols)
        hb = [capi.pinned_copy(x) for x in (b, n, disp)]
        outs = [capi.HostBuffer((1, ngroups * 16), np.int32) for _ in range(2)]
        best = None
        for rep in range(4):
            for o in outs:
                o.a[...] = -3
            ctx.wait()
            t0 = time.perf_counter()
            ctx.search_chunk_async(hb[0].a, hb[1].a, hb[2].a, outs[0].a, 16, dev=0)
            ctx.search_chunk_async(hb[0].a, hb[1].a, hb[2].a, outs[1].a, 16, dev=1)
            t_calls = time.perf_counter() - t0
            ctx.wait()
            t_all = time.perf_counter() - t0
            for o in outs:
                np.testing.assert_array_equal(o.a, want)
            if rep > 0 and (best is None or t_calls / t_all < best[0] / best[1]):   # (the first pass allocates the slots)
                best = (t_calls, t_all)
        assert best[0] < 0.5 * best[1], f"the two calls held the host {best[0] * 1e3:.2f} ms of the {best[1] * 1e3:.2f} ms until both devices were through"
        for x in hb + outs:
            x.close()


def test_one_call_search_keeps_a_converted_residue_buffer_alive(oracle):
    """oswald_hip_search_chunk_async reads `b` by DMA until the next wait.  The Python binding converts what it is
    given (np.ascontiguousarray): a caller's int16 or strided array becomes a temporary, which must live until wait() --
    as must the caller's own array when the caller drops it right after the call.  Several such calls in a row, each with
    its buffer dropped and the heap churned before the wait, must still give the oracle's scores."""
    import gc
    from oswald_amd import capi
    qs = synth.make_queries([60, 133], seed=41)
    L, R, O = random_db(1500, seed=42, min_len=30, max_len=300, queries=qs, homologs=2)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    disp = disp.astype(np.uint32)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        outs = []
        for form in ("int16", "strided", "dropped"):
            out = np.full((len(qs), len(n) * 16), -3, np.int32)
            if form == "int16":
                src = b.astype(np.int16)                       # a dtype conversion inside the binding
            elif form == "strided":
                src = np.repeat(b, 2)[::2]                     # not contiguous: a copy inside the binding
                assert not src.flags.c_contiguous
            else:
                src = b.copy()                                 # the caller's own array, dropped before the wait
            ctx.search_chunk_async(src, n, disp, out, 16)
            del src
            gc.collect()
            junk = [np.full(b.size, 0xEE, np.uint8) for _ in range(3)]   # whatever was freed is written over
            del junk
            outs.append(out)
        ctx.wait()
        for out in outs:
            np.testing.assert_array_equal(out[:, :want.shape[1]], want[:, :out.shape[1]])


def test_failures_come_back_as_error_codes_and_the_context_lives_on(oracle, monkeypatch):
    """(a) a device allocation that fails (OSWALD_HIP_FAIL_DEVICE_ALLOC_ABOVE: a device that is full -- here the
    buffers of the chunk slot the first upload needs; bring-up is spared by the hook) is OSWALD_HIP_ENOMEM, not a dead process, and the same context searches correctly once
    memory is there; (b) a host allocation that THROWS inside the library (a query buffer no memory can hold: std::length_error
    out of std::vector) comes back through the C ABI as OSWALD_HIP_ENOMEM too (every entry is guarded)."""
    import ctypes
    from oswald_amd import capi
    qs = synth.make_queries([80, 150, 222], seed=5)
    L, R, O = random_db(700, seed=6, min_len=30, max_len=260, queries=qs, homologs=2)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    disp = disp.astype(np.uint32)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = expect(oracle, qs, b, n, disp, 16, sm, 10, 2)
    monkeypatch.setenv("OSWALD_HIP_FAIL_DEVICE_ALLOC_ABOVE", str(64 << 10))
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        with pytest.raises(capi.OswaldHipError, match="error -4"):
            ctx.chunk_upload(b, n, disp, 16)                  # the slot's re-tiled residues (~250 KB) cannot be made
        monkeypatch.delenv("OSWALD_HIP_FAIL_DEVICE_ALLOC_ABOVE")
        ctx.set_scoring(sm, 10, 2)                            # (hooks are read when a context is configured)
        # (b): Q = 2^62 bytes of queries, no query: nothing is read, the copy of the buffer cannot be allocated
        rc = ctx.lib.oswald_hip_set_queries(ctx.h, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(1 << 62), None, None, 0)
        assert rc == -4 and "oswald_hip_set_queries" in ctx.lib.oswald_hip_last_error().decode()
        ctx.set_queries(a, m, ad)
        out = np.full((len(qs), len(n) * 16), -3, np.int32)
        h = ctx.chunk_upload(b, n, disp, 16)
        ctx.chunk_search(h, out)
        ctx.wait()
        ctx.chunk_release(h)
        np.testing.assert_array_equal(out[:, :want.shape[1]], want[:, :out.shape[1]])


def test_index_maps_given_in_a_row_do_not_overtake_their_readers(oracle):
    """a resident chunk searched three times, each time under another index map and without a wait in between.  The
    maps flip between two device buffers; the third map's copy must not land in the first map's buffer before the first search's
    top-list fold has read it.  Every search folds its chunk's best into the running list on DATABASE keys, so the final list
    holds each of the best sequences three times -- once per map, larger index first -- and a fold that ran on the wrong map would
    leave the first map's entries out."""
    from oswald_amd import capi
    qs = synth.make_queries([300, 420, 555], seed=71)
    L, R, O = random_db(6000, seed=72, min_len=200, max_len=900, queries=qs, homologs=2)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    disp = disp.astype(np.uint32)
    nseq = len(sl)
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        h = ctx.chunk_upload(b, n, disp, 16)
        table = np.zeros((len(qs), len(n) * 16), np.int32)
        ctx.chunk_search(h, table)
        ctx.wait()
        maps = [np.arange(nseq, dtype=np.uint32) + np.uint32(k * 1000000) for k in range(3)]
        for rep in range(3):
            ctx.topr_begin(6)
            for mp in maps:
                ctx.chunk_set_index(h, 0, nseq, mp)
                ctx.chunk_search(h, None)
            sc, ix = ctx.topr(6)
            for q in range(len(qs)):
                ws, wi = dblayout.topr_reference_order(table[q, :nseq], 2)
                want_s = np.repeat(ws, 3)
                want_i = (wi.astype(np.int64)[:, None] + np.array([2000000, 1000000, 0])[None, :]).reshape(-1)
                np.testing.assert_array_equal(sc[q], want_s)
                np.testing.assert_array_equal(ix[q].astype(np.int64), want_i)
        ctx.chunk_release(h)


def test_buffer_life_cycle_of_a_search(oracle):
    """ABI 5: the reference's buffer life cycle -- host buffers before the clock (FPGAsearch.c:69-74 -> oswald_hip_reserve_host), device
    buffers inside it (:85-96 -> oswald_hip_reserve_chunks), release behind it (:361-368 -> oswald_hip_release_chunks) -- three searches
    in a row in one context, each creating its device buffers anew; a chunk that is still held keeps its buffers through a release;
    scores and top lists are the oracle's every time."""
    from oswald_amd import capi
    qs = synth.make_queries([90, 137, 250, 301], seed=81)
    bfull, nfull, dfull, parts = _chunks(2500, qs, 150000, seed=82)
    assert len(parts) >= 4
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("life cycle", oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    nvalid = sum(p[4] for p in parts)
    big_b = max(p[0].size for p in parts)
    big_g = max(len(p[1]) for p in parts)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.reserve_host(big_g, 16, len(qs), 3)                      # before the "clock"
        held = None
        for rep in range(3):
            ctx.set_queries(a, m, ad)
            ctx.reserve_chunks(big_b, big_g, 16, len(qs), 3)         # inside it: the slots' device buffers
            ctx.topr_begin(5)
            outs = []
            for k, (b, n, disp, s0, nv) in enumerate(parts):
                h = ctx.chunk_upload(b, n, disp, 16, wait=False)
                ctx.chunk_set_index(h, s0, nv)
                out = np.full((len(qs), len(n) * 16), -3, np.int32)
                ctx.chunk_search(h, out)
                outs.append(out)
                if rep == 0 and k == 0:
                    held = (h, out.shape)                             # this chunk stays resident through the releases below
                else:
                    ctx.chunk_release(h)
            sc, ix = ctx.topr(5)
            ctx.wait()
            got = np.concatenate(outs, axis=1)
            np.testing.assert_array_equal(got[:, :want.shape[1]], want[:, :got.shape[1]])
            for q in range(len(qs)):
                ws, wi = dblayout.topr_reference_order(want[q, :nvalid], 5)
                np.testing.assert_array_equal(sc[q], ws)
                np.testing.assert_array_equal(ix[q], wi)
            ctx.release_chunks()                                      # behind it
            # the chunk that is still held was not touched: searched again, the same scores
            again = np.full(held[1], -3, np.int32)
            ctx.chunk_search(held[0], again)
            ctx.wait()
            np.testing.assert_array_equal(again, outs[0] if rep == 0 else first)
            if rep == 0:
                first = again.copy()
        ctx.chunk_release(held[0])
        ctx.release_chunks()


@pytest.mark.parametrize("tables", ["page-locked: written by the kernels", "pageable"])
def test_tails_through_the_pipeline(oracle, monkeypatch, tables):
    """The tails of SHORT pair items (every eligible item: OSWALD_HIP_PAIR_TAILS=2) on the paths the parity tests do not take: a
    first chunk cut in two by the library, searches queued behind uploads in flight, score tables in page-locked memory that the
    kernels write directly (the tail folds its best into what the pair part stored -- in both tables), slots re-used, the
    context-level top list on database keys."""
    from oswald_amd import capi
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "2")
    monkeypatch.setenv("OSWALD_HIP_SPLIT_BYTES", "30000")
    qs = synth.make_queries([70, 131, 188, 260, 340, 415], seed=91)          # three pairs, tails of 61 / 72 / 75 rows, no single query
    bfull, nfull, dfull, parts = _chunks(3000, qs, 200000, seed=92)
    assert len(parts) >= 4
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("tails pipeline", oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    nvalid = sum(p[4] for p in parts)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2)
        ctx.set_queries(a, m, ad)
        for rep in range(3):
            if tables.startswith("page-locked"):
                bufs = [capi.HostBuffer((len(qs), len(p[1]) * 16), np.int32) for p in parts]
                outs = [hb.a for hb in bufs]
            else:
                bufs, outs = [], [np.empty((len(qs), len(p[1]) * 16), np.int32) for p in parts]
            for o in outs:
                o[...] = -3
            ctx.topr_begin(7)
            hs = [ctx.chunk_upload(*parts[0][:3], 16, wait=False)]
            ctx.chunk_set_index(hs[0], parts[0][3], parts[0][4])
            for k in range(len(parts)):
                ctx.chunk_search(hs[k], outs[k])
                if k + 1 < len(parts):
                    hs.append(ctx.chunk_upload(*parts[k + 1][:3], 16, wait=False))
                    ctx.chunk_set_index(hs[k + 1], parts[k + 1][3], parts[k + 1][4])
                ctx.chunk_release(hs[k])
            sc, ix = ctx.topr(7)
            ctx.wait()
            got = np.concatenate([np.array(o) for o in outs], axis=1)
            np.testing.assert_array_equal(got[:, :want.shape[1]], want[:, :got.shape[1]])
            wsc, wix = _toplists(want, nvalid, 7)
            np.testing.assert_array_equal(sc, wsc)
            np.testing.assert_array_equal(ix.astype(np.int64), wix)
            for hb in bufs:
                hb.close()


@pytest.mark.parametrize("mode", ["int16, tails forced", "int16, padded pairs", "8-bit first pass", "int32 cells", "one query"])
def test_resident_chunks_searched_as_one_launch(oracle, monkeypatch, mode):
    """oswald_hip_search_resident: several resident chunks -- one of them cut in two by the library at its upload -- searched as ONE
    launch through a combined block table (column offsets counted from the lowest `tiled` address among the chunks, one score table).
    Score columns and the context-level top lists (first index for some members, an index map for one) are the oracle's; again with
    the group's tables and plan kept; again after one member has been released and uploaded anew (the group is made again); and the
    chunks searched one by one afterwards give the same."""
    from oswald_amd import capi
    monkeypatch.setenv("OSWALD_HIP_PAIRS", "2")
    monkeypatch.setenv("OSWALD_HIP_SPLIT_BYTES", "30000")
    monkeypatch.setenv("OSWALD_HIP_PAIR_TAILS", "0" if "padded" in mode else "2")
    bits = 8 if mode.startswith("8") else 32 if mode.startswith("int32") else 16
    qs = synth.make_queries([150] if mode == "one query" else [70, 131, 188, 260, 340, 415], seed=95)
    bfull, nfull, dfull, parts = _chunks(3200, qs, 220000, seed=96)
    assert len(parts) >= 4
    sm = submat.load("blosum62")
    a, m, ad = pack_queries(qs)
    want = _want("resident " + ("1" if len(qs) == 1 else "6"), oracle, qs, bfull, nfull, dfull, sm, 10, 2)
    nvalid = sum(p[4] for p in parts)
    rng = np.random.default_rng(3)
    with capi.Context(1) as ctx:
        ctx.set_scoring(sm, 10, 2, bits)
        ctx.set_queries(a, m, ad)
        hs = []
        for k, (b, n, disp, s0, nv) in enumerate(parts):
            hs.append(ctx.chunk_upload(b, n, disp, 16, wait=(k != 0)))      # the first one asynchronously on an idle device: cut in two
        ctx.wait()

        def index_all():
            for k, (b, n, disp, s0, nv) in enumerate(parts):
                ctx.chunk_set_index(hs[k], s0, nv, (np.arange(nv, dtype=np.uint32) + s0) if k == 1 else None)

        def check(out, sc, ix):
            np.testing.assert_array_equal(out[:, :want.shape[1]], want[:, :out.shape[1]])
            wsc, wix = _toplists(want, nvalid, 9)
            np.testing.assert_array_equal(sc, wsc)
            np.testing.assert_array_equal(ix.astype(np.int64), wix)

        for rep in range(4):
            if rep == 2:   # one member leaves and comes back (another upload: the group must be made again)
                ctx.chunk_release(hs[2])
                hs[2] = ctx.chunk_upload(*parts[2][:3], 16)
            ctx.topr_begin(9)
            index_all()
            out = np.full((len(qs), sum(len(p[1]) for p in parts) * 16), -3, np.int32)
            ctx.search_resident(hs, out)
            sc, ix = ctx.topr(9)
            ctx.wait()
            check(out, sc, ix)
        # ... and one by one
        ctx.topr_begin(9)
        index_all()
        outs = [np.full((len(qs), len(p[1]) * 16), -3, np.int32) for p in parts]
        for k in range(len(parts)):
            ctx.chunk_search(hs[k], outs[k])
        sc, ix = ctx.topr(9)
        ctx.wait()
        check(np.concatenate(outs, axis=1), sc, ix)
        # a handle given twice, an empty list: errors, not faults
        with pytest.raises(capi.OswaldHipError):
            ctx.search_resident([hs[0], hs[0]])
        for h in hs:
            ctx.chunk_release(h)
