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
    assert capi.load().oswald_hip_abi_version() == 1


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
