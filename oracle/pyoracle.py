"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes door onto oracle/liboswald_oracle.so (the CPU restatement in
sw_oracle.c).  Imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- as the checker / the reported baseline, never
by anything under oswald_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboswald_oracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libosw_ref.so")

_lib = None


def build(ref: bool = False):
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["ref"] if ref else []))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.osw_oracle_sw_scalar.restype = C.c_int32
        lib.osw_oracle_sw_lane_sat.restype = C.c_int32
        lib.osw_oracle_search_chunk_simd.restype = C.c_int
        lib.osw_oracle_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def alphabet_map(raw: bytes | np.ndarray) -> np.ndarray:
    src = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray)) else np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.empty_like(src)
    load().osw_oracle_alphabet_map(_p(src), _p(out), C.c_uint64(src.size))
    return out


def sw_scalar(a, b, submat, open_gap, extend_gap, stride=1, n=None) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    if n is None:
        n = b.size // stride
    return int(load().osw_oracle_sw_scalar(_p(a), C.c_uint32(a.size), _p(b), C.c_uint32(n), C.c_uint32(stride), _p(sm), open_gap, extend_gap))


def sw_lane_sat(a, b, submat, open_gap, extend_gap, bits, stride=1, n=None) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    if n is None:
        n = b.size // stride
    return int(load().osw_oracle_sw_lane_sat(_p(a), C.c_uint32(a.size), _p(b), C.c_uint32(n), C.c_uint32(stride), _p(sm), open_gap, extend_gap, bits))


def group_escalate(a, b_group, n, W, submat, open_gap, extend_gap):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b_group, dtype=np.uint8)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    sc = np.zeros(W, dtype=np.int32)
    st = np.zeros(W, dtype=np.int32)
    load().osw_oracle_group_escalate(_p(a), C.c_uint32(a.size), _p(b), C.c_uint32(n), C.c_uint32(W), _p(sm), open_gap, extend_gap, _p(sc), _p(st))
    return sc, st


def _chunk_args(a, m, a_disp, b, n, disp, submat):
    return (np.ascontiguousarray(a, dtype=np.uint8), np.ascontiguousarray(m, dtype=np.uint16),
            np.ascontiguousarray(a_disp, dtype=np.uint32), np.ascontiguousarray(b, dtype=np.uint8),
            np.ascontiguousarray(n, dtype=np.uint16), np.ascontiguousarray(disp, dtype=np.uint32),
            np.ascontiguousarray(submat, dtype=np.int8))


def search_chunk_scalar(a, m, a_disp, b, n, disp, W, submat, open_gap, extend_gap, threads=0) -> np.ndarray:
    """Exact scores [nq][ngroups*W] by the scalar restatement."""
    a, m, a_disp, b, n, disp, sm = _chunk_args(a, m, a_disp, b, n, disp, submat)
    out = np.zeros((m.size, n.size * W), dtype=np.int32)
    if threads <= 0:
        threads = load().osw_oracle_max_threads()
    load().osw_oracle_search_chunk_scalar(_p(a), _p(m), _p(a_disp), C.c_uint32(m.size), _p(b), _p(n), _p(disp), C.c_uint32(n.size),
                                          C.c_uint32(W), _p(sm), open_gap, extend_gap, _p(out), threads)
    return out


def search_chunk_simd(a, m, a_disp, b, n, disp, W, submat, open_gap, extend_gap, block=256, threads=0):
    """The SSE4.1 (W=16) / AVX2 (W=32) port of the host path.  Returns
    (scores [nq][ngroups*W], cells computed per precision [int8, int16, int32])."""
    a, m, a_disp, b, n, disp, sm = _chunk_args(a, m, a_disp, b, n, disp, submat)
    out = np.zeros((m.size, n.size * W), dtype=np.int32)
    stage = np.zeros(3, dtype=np.uint64)
    if threads <= 0:
        threads = load().osw_oracle_max_threads()
    rc = load().osw_oracle_search_chunk_simd(_p(a), _p(m), _p(a_disp), C.c_uint32(m.size), _p(b), _p(n), _p(disp), C.c_uint32(n.size),
                                             C.c_uint32(W), _p(sm), open_gap, extend_gap, C.c_uint32(block), threads, _p(out), _p(stage))
    if rc != 0:
        raise ValueError("unsupported lane width")
    return out, stage


def sort_scores(scores):
    sc = np.ascontiguousarray(scores, dtype=np.int32).copy()
    ix = np.zeros(sc.size, dtype=np.uint32)
    load().osw_oracle_sort_scores(_p(sc), _p(ix), C.c_uint64(sc.size))
    return sc, ix


def max_threads() -> int:
    return int(load().osw_oracle_max_threads())
