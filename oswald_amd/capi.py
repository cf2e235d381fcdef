"""ctypes binding of liboswald_hip.so (include/oswald_hip.h).

This module is plumbing only: it loads the in-tree shared library built by
`__graft_entry__.build()` / `make -C oswald_amd/csrc` and exposes the C ABI.
There is no fallback: if the library is missing, or no GPU is present at
`Context()` time, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboswald_hip.so")
# tools/ only: OSWALD_HIP_USE_DIAG_LIB=1 loads the -DOSW_DIAG build (`make -C oswald_amd/csrc diag`), which also reads the
# planner sweep knobs and the kernel timing diagnostics from the environment; tests and bench.py's records never set it
if os.environ.get("OSWALD_HIP_USE_DIAG_LIB"):
    LIB_PATH = os.path.join(_HERE, "liboswald_hip_diag.so" if os.environ["OSWALD_HIP_USE_DIAG_LIB"] == "1" else os.environ["OSWALD_HIP_USE_DIAG_LIB"])  # (or another build's file name, for A/B runs)

# every symbol include/oswald_hip.h declares
SYMBOLS = (
    "oswald_hip_abi_version", "oswald_hip_last_error", "oswald_hip_device_count", "oswald_hip_init", "oswald_hip_finalize",
    "oswald_hip_info", "oswald_hip_set_scoring", "oswald_hip_set_queries", "oswald_hip_chunk_upload", "oswald_hip_chunk_search",
    "oswald_hip_chunk_release", "oswald_hip_search_chunk_async", "oswald_hip_wait", "oswald_hip_chunk_topr",
    "oswald_hip_set_profiling", "oswald_hip_kernel_stats", "oswald_hip_chunk_geometry", "oswald_hip_chunk_upload_async", "oswald_hip_reserve", "oswald_hip_rerun_counts",
    "oswald_hip_topr_begin", "oswald_hip_chunk_set_index", "oswald_hip_topr", "oswald_hip_merge_candidates",
    "oswald_hip_comm_unique_id", "oswald_hip_comm_init_rank", "oswald_hip_comm_info", "oswald_hip_max_chunk_size",
    "oswald_hip_host_alloc", "oswald_hip_host_free", "oswald_hip_rerun_stats",
    "oswald_hip_host_register", "oswald_hip_host_unregister", "oswald_hip_comm_destroy", "oswald_hip_reserve_chunks",
    "oswald_hip_chunk_wait", "oswald_hip_release_chunks", "oswald_hip_reserve_host", "oswald_hip_search_resident",
)
COMM_ID_BYTES = 128   # OSWALD_HIP_COMM_ID_BYTES


class OswaldHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load liboswald_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OswaldHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_size_t
    lib.oswald_hip_abi_version.restype = i32
    lib.oswald_hip_last_error.restype = C.c_char_p
    lib.oswald_hip_device_count.argtypes = [C.POINTER(i32)]
    lib.oswald_hip_init.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    lib.oswald_hip_finalize.argtypes = [vp]
    lib.oswald_hip_info.argtypes = [vp, i32, C.c_char_p, sz]
    lib.oswald_hip_set_scoring.argtypes = [vp, vp, i32, i32, i32]
    lib.oswald_hip_set_queries.argtypes = [vp, vp, u64, vp, vp, u32]
    lib.oswald_hip_chunk_upload.argtypes = [vp, i32, vp, u64, vp, vp, u32, u32, C.POINTER(i32)]
    lib.oswald_hip_chunk_upload_async.argtypes = [vp, i32, vp, u64, vp, vp, u32, u32, C.POINTER(i32)]
    lib.oswald_hip_reserve.argtypes = [vp, i32, u32]
    lib.oswald_hip_reserve_chunks.argtypes = [vp, i32, u64, u32, u32, u32, u32]
    lib.oswald_hip_release_chunks.argtypes = [vp, i32]
    lib.oswald_hip_search_resident.argtypes = [vp, i32, C.POINTER(i32), u32, vp]
    lib.oswald_hip_reserve_host.argtypes = [vp, i32, u32, u32, u32, u32]
    lib.oswald_hip_rerun_counts.argtypes = [vp, i32, C.POINTER(u64)]
    lib.oswald_hip_chunk_search.argtypes = [vp, i32, i32, vp]
    lib.oswald_hip_chunk_release.argtypes = [vp, i32, i32]
    lib.oswald_hip_chunk_wait.argtypes = [vp, i32, i32]
    lib.oswald_hip_search_chunk_async.argtypes = [vp, i32, vp, u64, vp, vp, u32, u32, vp]
    lib.oswald_hip_wait.argtypes = [vp, i32]
    lib.oswald_hip_chunk_topr.argtypes = [vp, i32, i32, u32, u32, vp, vp]
    lib.oswald_hip_topr_begin.argtypes = [vp, u32]
    lib.oswald_hip_chunk_set_index.argtypes = [vp, i32, i32, u32, u32, vp]
    lib.oswald_hip_topr.argtypes = [vp, u32, vp, vp]
    lib.oswald_hip_merge_candidates.argtypes = [u32, u64, vp, vp, u32, vp, vp]
    lib.oswald_hip_rerun_stats.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.oswald_hip_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.oswald_hip_host_free.argtypes = [vp]
    lib.oswald_hip_host_register.argtypes = [vp, sz]
    lib.oswald_hip_host_unregister.argtypes = [vp]
    lib.oswald_hip_comm_destroy.argtypes = [vp]
    lib.oswald_hip_max_chunk_size.argtypes = [vp, i32, u32, u32, C.POINTER(u64)]
    lib.oswald_hip_comm_unique_id.argtypes = [vp, sz]
    lib.oswald_hip_comm_init_rank.argtypes = [vp, vp, sz, i32, i32]
    lib.oswald_hip_comm_info.argtypes = [vp, C.POINTER(i32)]
    lib.oswald_hip_set_profiling.argtypes = [vp, i32]
    lib.oswald_hip_kernel_stats.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64), i32]
    lib.oswald_hip_chunk_geometry.argtypes = [vp, i32, i32, C.POINTER(u64)]
    for name in SYMBOLS:
        if name != "oswald_hip_last_error":
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc):
    if rc != 0:
        raise OswaldHipError(f"liboswald_hip: error {rc}: {load().oswald_hip_last_error().decode(errors='replace')}")


def merge_candidates(cand_scores: np.ndarray, cand_index: np.ndarray, r: int):
    """oswald_hip_merge_candidates: [nq][K] candidates (score < 0 = empty slot) -> the r best per query in the
    reference's order (descending score, ties by descending database index): ([nq][r] int32, [nq][r] uint32),
    empty slots (-1, 0xffffffff).  Host logic of the library (no GPU needed)."""
    cs = np.ascontiguousarray(cand_scores, dtype=np.int32)
    ci = np.ascontiguousarray(cand_index, dtype=np.uint32)
    assert cs.ndim == 2 and cs.shape == ci.shape
    out_s = np.empty((cs.shape[0], r), np.int32)
    out_i = np.empty((cs.shape[0], r), np.uint32)
    _chk(load().oswald_hip_merge_candidates(cs.shape[0], cs.shape[1], _ptr(cs), _ptr(ci), r, _ptr(out_s), _ptr(out_i)))
    return out_s, out_i


class HostBuffer:
    """Page-locked host memory from oswald_hip_host_alloc as a numpy array (`.a`); freed by close() / garbage collection."""

    def __init__(self, shape, dtype):
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        self.p = C.c_void_p()
        _chk(load().oswald_hip_host_alloc(max(nbytes, 1), C.byref(self.p)))
        self.a = np.frombuffer((C.c_char * max(nbytes, 1)).from_address(self.p.value), dtype=dtype, count=nbytes // np.dtype(dtype).itemsize).reshape(shape)

    def close(self):
        if self.p:
            self.a = None
            load().oswald_hip_host_free(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def pinned_copy(x: np.ndarray) -> HostBuffer:
    hb = HostBuffer(x.shape, x.dtype)
    hb.a[...] = x
    return hb


class Registered:
    """A numpy array page-locked in place (oswald_hip_host_register) for as long as this object lives: uploads from it are
    asynchronous DMA.  The array must not be resized or freed meanwhile (a reference is kept)."""

    def __init__(self, a: np.ndarray):
        assert a.flags.c_contiguous
        self.a = a
        self.p = C.c_void_p(a.ctypes.data)
        _chk(load().oswald_hip_host_register(self.p, a.nbytes))

    def close(self):
        if self.p:
            load().oswald_hip_host_unregister(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def comm_unique_id() -> bytes:
    """oswald_hip_comm_unique_id: the id rank 0 hands to the other ranks (COMM_ID_BYTES bytes)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(load().oswald_hip_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def device_count() -> int:
    n = C.c_int(0)
    rc = load().oswald_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    """One oswald_hip_ctx: the operation sequence of the reference's search
    driver (reference host/src/FPGAsearch.c:82-238) through the C ABI."""

    def __init__(self, ndev: int = 1, device_ids=None):
        self.lib = load()
        self.h = C.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (C.c_int * ndev)(*device_ids)
        _chk(self.lib.oswald_hip_init(ndev, ids, C.byref(self.h)))
        self.ndev = ndev
        self.nq = 0
        self._keep = []

    def close(self):
        if self.h:
            self.lib.oswald_hip_finalize(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self, dev: int = 0) -> str:
        buf = C.create_string_buffer(2048)
        _chk(self.lib.oswald_hip_info(self.h, dev, buf, len(buf)))
        return buf.value.decode()

    def set_scoring(self, submat: np.ndarray, open_gap: int, extend_gap: int, cell_bits: int = 0):
        sm = np.ascontiguousarray(submat, dtype=np.int8).reshape(-1)
        assert sm.size == 24 * 32
        _chk(self.lib.oswald_hip_set_scoring(self.h, _ptr(sm), open_gap, extend_gap, cell_bits))

    def set_queries(self, a: np.ndarray, m: np.ndarray, a_disp: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        m = np.ascontiguousarray(m, dtype=np.uint16)
        a_disp = np.ascontiguousarray(a_disp, dtype=np.uint32)
        _chk(self.lib.oswald_hip_set_queries(self.h, _ptr(a), a.size, _ptr(m), _ptr(a_disp), m.size))
        self.nq = int(m.size)

    def chunk_upload(self, b, n, disp, lane_width: int = 16, dev: int = 0, wait: bool = True) -> int:
        """wait=False: oswald_hip_chunk_upload_async (the arrays are kept alive until wait())."""
        b = np.ascontiguousarray(b, dtype=np.uint8)
        n = np.ascontiguousarray(n, dtype=np.uint16)
        disp = np.ascontiguousarray(disp, dtype=np.uint32)
        h = C.c_int(-1)
        if wait:
            _chk(self.lib.oswald_hip_chunk_upload(self.h, dev, _ptr(b), b.size, _ptr(n), _ptr(disp), n.size, lane_width, C.byref(h)))
        else:
            self._keep += [b, n, disp]
            _chk(self.lib.oswald_hip_chunk_upload_async(self.h, dev, _ptr(b), b.size, _ptr(n), _ptr(disp), n.size, lane_width, C.byref(h)))
        return h.value

    def reserve(self, max_sequence_length: int, dev: int = -1):
        _chk(self.lib.oswald_hip_reserve(self.h, dev, max_sequence_length))

    def reserve_chunks(self, chunk_bytes: int, ngroups: int, lane_width: int = 16, nq: int = 0, slots: int = 3, dev: int = -1):
        _chk(self.lib.oswald_hip_reserve_chunks(self.h, dev, chunk_bytes, ngroups, lane_width, nq, slots))

    def reserve_host(self, ngroups: int, lane_width: int = 16, nq: int = 0, slots: int = 3, dev: int = -1):
        """The slots' page-locked host staging only (before a caller's clock; reserve_chunks inside it makes the device buffers)."""
        _chk(self.lib.oswald_hip_reserve_host(self.h, dev, ngroups, lane_width, nq, slots))

    def release_chunks(self, dev: int = -1):
        """Gives the buffers of every chunk slot that holds no chunk back (waits for the device)."""
        _chk(self.lib.oswald_hip_release_chunks(self.h, dev))

    def chunk_search(self, chunk: int, out: np.ndarray | None = None, dev: int = 0):
        """Asynchronous; `out` (int32 [nq][ngroups*W]) is valid after wait()."""
        if out is not None:
            assert out.dtype == np.int32 and out.flags.c_contiguous
            self._keep.append(out)
        _chk(self.lib.oswald_hip_chunk_search(self.h, dev, chunk, _ptr(out)))

    def search_resident(self, chunks, out: np.ndarray | None = None, dev: int = 0):
        """All queries against several resident chunks as ONE launch; `out` (int32 [nq][sum of the chunks' ngroups*W]) is valid after wait()."""
        hs = (C.c_int * len(chunks))(*[int(h) for h in chunks])
        if out is not None:
            assert out.dtype == np.int32 and out.flags.c_contiguous
            self._keep.append(out)
        _chk(self.lib.oswald_hip_search_resident(self.h, dev, hs, len(chunks), _ptr(out)))

    def chunk_wait(self, chunk: int, dev: int = 0):
        """Blocks until the chunk's last search is through (whatever has been queued behind it)."""
        _chk(self.lib.oswald_hip_chunk_wait(self.h, dev, chunk))

    def chunk_release(self, chunk: int, dev: int = 0):
        _chk(self.lib.oswald_hip_chunk_release(self.h, dev, chunk))

    def search_chunk_async(self, b, n, disp, out: np.ndarray, lane_width: int = 16, dev: int = 0):
        b = np.ascontiguousarray(b, dtype=np.uint8)
        n = np.ascontiguousarray(n, dtype=np.uint16)
        disp = np.ascontiguousarray(disp, dtype=np.uint32)
        assert out.dtype == np.int32 and out.flags.c_contiguous
        # the call is asynchronous throughout: `b` is read by DMA until wait() (include/oswald_hip.h); ascontiguousarray may
        # have made a temporary of it.  n and disp are copied by the library inside the call.
        self._keep += [b, out]
        _chk(self.lib.oswald_hip_search_chunk_async(self.h, dev, _ptr(b), b.size, _ptr(n), _ptr(disp), n.size, lane_width, _ptr(out)))

    def wait(self, dev: int = -1):
        _chk(self.lib.oswald_hip_wait(self.h, dev))
        self._keep.clear()

    def chunk_topr(self, chunk: int, nvalid: int, r: int, dev: int = 0):
        sc = np.empty((self.nq, r), dtype=np.int32)
        ix = np.empty((self.nq, r), dtype=np.uint32)
        _chk(self.lib.oswald_hip_chunk_topr(self.h, dev, chunk, nvalid, r, _ptr(sc), _ptr(ix)))
        return sc, ix

    def topr_begin(self, r: int):
        _chk(self.lib.oswald_hip_topr_begin(self.h, r))

    def chunk_set_index(self, chunk: int, first_index: int, nvalid: int, index_map=None, dev: int = 0):
        im = None if index_map is None else np.ascontiguousarray(index_map, dtype=np.uint32)
        assert im is None or im.size >= nvalid
        _chk(self.lib.oswald_hip_chunk_set_index(self.h, dev, chunk, first_index, nvalid, _ptr(im)))

    def topr(self, r: int):
        """([nq][r] int32 scores, [nq][r] uint32 database indices) over everything searched since topr_begin."""
        sc = np.empty((self.nq, r), dtype=np.int32)
        ix = np.empty((self.nq, r), dtype=np.uint32)
        _chk(self.lib.oswald_hip_topr(self.h, r, _ptr(sc), _ptr(ix)))
        return sc, ix

    def max_chunk_size(self, nq: int, max_sequence_length: int, dev: int = 0) -> int:
        out = C.c_uint64(0)
        _chk(self.lib.oswald_hip_max_chunk_size(self.h, dev, nq, max_sequence_length, C.byref(out)))
        return int(out.value)

    def comm_init_rank(self, comm_id: bytes, nranks: int, rank: int):
        """Collective: joins the process-level RCCL communicator; topr() then returns the list of all ranks."""
        assert len(comm_id) >= COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(comm_id), len(comm_id))
        _chk(self.lib.oswald_hip_comm_init_rank(self.h, buf, len(comm_id), nranks, rank))

    def comm_destroy(self):
        """Gives the process-level communicator up again (every rank, before the job takes another way)."""
        _chk(self.lib.oswald_hip_comm_destroy(self.h))

    def comm_info(self):
        out = (C.c_int * 4)()
        _chk(self.lib.oswald_hip_comm_info(self.h, out))
        return {"context_ranks": int(out[0]), "process_ranks": int(out[1]), "process_rank": int(out[2]), "rccl_version": int(out[3])}

    def set_profiling(self, on: bool):
        _chk(self.lib.oswald_hip_set_profiling(self.h, 1 if on else 0))

    def kernel_stats(self, dev: int = 0, reset: bool = False):
        ms, n, re = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _chk(self.lib.oswald_hip_kernel_stats(self.h, dev, C.byref(ms), C.byref(n), C.byref(re), 1 if reset else 0))
        return ms.value, n.value, re.value

    def rerun_stats(self, dev: int = 0):
        """(ms in the int16 re-run launches of the 8-bit pass, ms in the int32 re-run launches) since the last stats reset."""
        out = (C.c_double * 2)()
        _chk(self.lib.oswald_hip_rerun_stats(self.h, dev, out))
        return float(out[0]), float(out[1])

    def rerun_counts(self, dev: int = 0):
        """(items the 8-bit pass sent to the int16 re-run, sequences the int16 cells sent to the int32 re-run)."""
        out = (C.c_uint64 * 2)()
        _chk(self.lib.oswald_hip_rerun_counts(self.h, dev, out))
        return int(out[0]), int(out[1])

    def chunk_geometry(self, chunk: int, dev: int = 0):
        out = (C.c_uint64 * 8)()
        _chk(self.lib.oswald_hip_chunk_geometry(self.h, dev, chunk, out))
        return {"blocks": out[0], "col4_stored": out[1], "col4_live": out[2], "residue_bytes_per_query": out[3],
                "work_items": out[4], "max_log2_geometry": out[5], "planned_spill_bytes": out[6]}
