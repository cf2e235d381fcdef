"""The C-ABI library loads in a GPU-less container, exports every symbol that
include/oswald_hip.h declares, and refuses to work without a GPU (no CPU
fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

from oswald_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "oswald_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oswald_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert capi.load().oswald_hip_abi_version() == 5


def test_library_links_rccl():
    """The multi-GPU top-r gather runs over RCCL inside the C ABI: the library itself links librccl."""
    import subprocess
    out = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert re.search(r"NEEDED.*librccl\.so", out), out


def test_no_gpu_means_loud_failure():
    """Where no GPU is visible the layer must fail with a message, never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert capi.device_count() == 0
    with pytest.raises(capi.OswaldHipError) as e:
        capi.Context(1)
    assert "no CPU path" in str(e.value) or "hip" in str(e.value).lower()


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under oswald_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oswald_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"pyoracle|sw_oracle|liboswald_oracle|osw_oracle_|oracle/", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_header_documents_the_cell_modes():
    """cell_bits 0 / 16 / 32 are part of the ABI contract (include/oswald_hip.h)."""
    text = open(os.path.join(ROOT, "include", "oswald_hip.h")).read()
    for word in ("cell_bits", "packed int16", "22256", "int32"):
        assert word in text


def test_merge_candidates_is_the_reference_order():
    """oswald_hip_merge_candidates (the library's one top-list merge; host logic, runs without a GPU) against the
    numpy mirror of sort_scores()'s order (reference utils.c:3-86: descending score, ties by DESCENDING index) --
    many ties, empty slots, fewer candidates than r."""
    import numpy as np
    from oswald_amd import dblayout
    rng = np.random.default_rng(5)
    for nq, K, r in ((1, 1, 1), (3, 40, 10), (5, 300, 25), (2, 7, 12), (4, 0, 3)):
        sc = rng.integers(0, 6, size=(nq, K)).astype(np.int32)
        ix = np.stack([rng.permutation(100000)[:K] for _ in range(nq)]).astype(np.int64) if K else np.zeros((nq, 0), np.int64)
        empty = rng.random((nq, K)) < 0.2
        ix[empty] = -1
        got_s, got_i = capi.merge_candidates(np.where(empty, -1, sc), np.where(empty, 0, ix).astype(np.uint32), r)
        want_s, want_i = dblayout.merge_topr_rows(sc, ix, r)
        np.testing.assert_array_equal(got_s, want_s)
        np.testing.assert_array_equal(np.where(got_s < 0, -1, got_i.astype(np.int64)), want_i)
        assert ((got_i == 0xFFFFFFFF) == (got_s < 0)).all()


def test_no_exception_crosses_the_c_abi():
    """an allocation that throws inside the library (std::length_error / std::bad_alloc from a standard
    container) comes back as OSWALD_HIP_ENOMEM with the entry's name in oswald_hip_last_error() -- it used to be
    std::terminate, i.e. a dead caller.  oswald_hip_merge_candidates is host logic and reserves room for its candidates
    before it reads any: a candidate count no memory can hold fails there, in this process, and the process lives on."""
    import numpy as np
    lib = capi.load()
    cs = np.zeros(16, np.int32)
    ci = np.zeros(16, np.uint32)
    out_s = np.zeros(10, np.int32)
    out_i = np.zeros(10, np.uint32)
    for ncand in (1 << 61, (1 << 64) - 1):
        rc = lib.oswald_hip_merge_candidates(1, ctypes.c_uint64(ncand), cs.ctypes.data_as(ctypes.c_void_p), ci.ctypes.data_as(ctypes.c_void_p), 10,
                                             out_s.ctypes.data_as(ctypes.c_void_p), out_i.ctypes.data_as(ctypes.c_void_p))
        assert rc == -4, rc                                             # OSWALD_HIP_ENOMEM
        assert "oswald_hip_merge_candidates" in lib.oswald_hip_last_error().decode()
    # ... and the library still works
    got_s, got_i = capi.merge_candidates(np.array([[5, 7, 7]], np.int32), np.array([[1, 2, 3]], np.uint32), 2)
    assert got_s.tolist() == [[7, 7]] and got_i.tolist() == [[3, 2]]


def test_every_exported_entry_is_guarded():
    """Every extern "C" entry of oswald_hip.cpp that can fail is its implementation inside guarded() (the catch-all)."""
    src = open(os.path.join(ROOT, "oswald_amd", "csrc", "oswald_hip.cpp")).read()
    exported = re.findall(r"^int (oswald_hip_\w+)\(", src, flags=re.M)
    guarded = re.findall(r'return guarded\("(oswald_hip_\w+)"', src)
    assert sorted(exported) == sorted(set(guarded) | {"oswald_hip_abi_version"})
    assert set(capi.SYMBOLS) == set(exported) | {"oswald_hip_last_error"}
    assert "noexcept" in src[src.index("int guarded("):src.index("int guarded(") + 80]
