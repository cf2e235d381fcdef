"""ctypes door onto liboswald_host.so (the C++ host side, oswald_amd/host/)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oswald_amd", "liboswald_host.so")
CLI = os.path.join(ROOT, "oswald_amd", "oswald")

_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(PATH)
        lib.oswald_host_last_error.restype = C.c_char_p
        lib.oswald_host_submat.restype = C.c_void_p
        lib.oswald_host_encode.restype = C.c_uint8
        lib.oswald_host_encode.argtypes = [C.c_uint8]
        for f in ("oswald_host_queries_a", "oswald_host_queries_m", "oswald_host_queries_disp", "oswald_host_chunk_b",
                  "oswald_host_chunk_n", "oswald_host_chunk_nbb", "oswald_host_chunk_disp"):
            getattr(lib, f).restype = C.c_void_p
        lib.oswald_host_queries_title.restype = C.c_char_p
        lib.oswald_host_queries_title.argtypes = [C.c_uint64]
        lib.oswald_host_header.restype = C.c_char_p
        lib.oswald_host_header.argtypes = [C.c_uint64]
        for f in ("oswald_host_chunk_groups", "oswald_host_chunk_accum", "oswald_host_chunk_vD"):
            getattr(lib, f).restype = C.c_uint64
        _lib = lib
    return _lib


def _arr(ptr, ctype, n):
    if n == 0:
        return np.zeros(0, dtype=np.dtype(ctype))
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).copy()


def preprocess(fasta, out, threads=1):
    st = (C.c_uint64 * 3)()
    rc = load().oswald_host_preprocess(fasta.encode(), out.encode(), threads, st)
    if rc:
        raise RuntimeError(load().oswald_host_last_error().decode())
    return int(st[0]), int(st[1]), int(st[2])


def load_queries(fasta):
    lib = load()
    nq, Q = C.c_uint64(0), C.c_uint64(0)
    if lib.oswald_host_load_queries(fasta.encode(), C.byref(nq), C.byref(Q)):
        raise RuntimeError(lib.oswald_host_last_error().decode())
    return {"a": _arr(lib.oswald_host_queries_a(), C.c_uint8, Q.value), "m": _arr(lib.oswald_host_queries_m(), C.c_uint16, nq.value),
            "disp": _arr(lib.oswald_host_queries_disp(), C.c_uint32, nq.value + 1),
            "titles": [lib.oswald_host_queries_title(i) for i in range(nq.value)]}


def assemble(dbname, W=16, max_chunk=134217728, ndev=1):
    lib = load()
    out8 = (C.c_uint64 * 8)()
    if lib.oswald_host_assemble(dbname.encode(), W, C.c_uint64(max_chunk), ndev, out8):
        raise RuntimeError(lib.oswald_host_last_error().decode())
    keys = ("seqs", "D", "maxlen", "maxtitle", "vgroups", "vD", "max_chunk_vD", "chunk_count")
    r = {k: int(out8[i]) for i, k in enumerate(keys)}
    r["chunks"] = []
    for c in range(r["chunk_count"]):
        g = lib.oswald_host_chunk_groups(c)
        vD = lib.oswald_host_chunk_vD(c)
        r["chunks"].append({"groups": int(g), "accum": int(lib.oswald_host_chunk_accum(c)), "vD": int(vD),
                            "n": _arr(lib.oswald_host_chunk_n(c), C.c_uint16, g), "nbb": _arr(lib.oswald_host_chunk_nbb(c), C.c_uint16, g),
                            "disp": _arr(lib.oswald_host_chunk_disp(c), C.c_uint32, g), "b": _arr(lib.oswald_host_chunk_b(c), C.c_uint8, vD)})
    return r


def host_search_chunk(chunk, nq, groups, matrix, go, ge, threads=4, vector_length=32, block_width=256):
    """oswald::host_search_groups (the `-m 2` / hybrid host kernel) on a chunk of the database assembled last, for
    the queries loaded last: int32 [nq][groups*16].  vector_length: the command line's -v (16: SSE4.1 kernel, 32: AVX2);
    block_width: its -b (query rows per block of the 8-bit stage; 0: unblocked)."""
    out = np.zeros((nq, groups * 16), np.int32)
    if load().oswald_host_search_chunk_vb(chunk, matrix.encode(), go, ge, threads, vector_length, block_width, out.ctypes.data_as(C.c_void_p)):
        raise RuntimeError(load().oswald_host_last_error().decode())
    return out


def from_cache():
    """Did the last assemble() take its chunks from <db>.g16?"""
    return bool(load().oswald_host_db_from_cache())


def write_group_cache(dbname):
    if load().oswald_host_write_group_cache(dbname.encode()):
        raise RuntimeError(load().oswald_host_last_error().decode())


def headers_at(dbname, indices):
    lib = load()
    ix = np.ascontiguousarray(indices, np.uint64)
    if lib.oswald_host_load_headers_at(dbname.encode(), ix.ctypes.data_as(C.c_void_p), C.c_uint64(ix.size)):
        raise RuntimeError(lib.oswald_host_last_error().decode())
    return [lib.oswald_host_header(i) for i in range(ix.size)]


def headers(dbname, count):
    lib = load()
    if lib.oswald_host_load_headers(dbname.encode(), C.c_uint64(count)):
        raise RuntimeError(lib.oswald_host_last_error().decode())
    return [lib.oswald_host_header(i) for i in range(count)]


def top_scores(scores, r):
    sc = np.ascontiguousarray(scores, np.int32)
    r = min(r, sc.size)
    os_ = np.zeros(r, np.int32)
    oi = np.zeros(r, np.uint64)
    load().oswald_host_top_scores(sc.ctypes.data_as(C.c_void_p), C.c_uint64(sc.size), C.c_uint64(r), os_.ctypes.data_as(C.c_void_p),
                                  oi.ctypes.data_as(C.c_void_p))
    return os_, oi


def submat(name):
    p = load().oswald_host_submat(name.encode())
    return None if not p else _arr(p, C.c_int8, 768)
