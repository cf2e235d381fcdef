"""ISA check of the packed-int16 kernels (CPU test: hipcc cross-compiles without a GPU).  The checks themselves live in
tools/isa_check.py (see there for what the hand-scheduled column loop needs from the compiler); `__graft_entry__.build()`
runs them too and stamps the library it built, and tests/test_gpu_isa_guard.py checks on the GPU box that the library
the process loaded is the stamped one."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_check.py"))
isa_check = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_check)


@pytest.fixture(scope="module")
def isa():
    if not isa_check.hipcc_path():
        pytest.skip("hipcc not available")
    return isa_check.compile_to_asm()


def test_compiler_never_touches_the_inflight_registers(isa):
    seen_asm_use, bad = isa_check.check_inflight_registers(isa)
    assert seen_asm_use > 1000, "the asm blocks that use the fixed registers were not found"
    assert not bad, "compiler-scheduled instructions touch in-flight registers: %r" % bad[:8]


def test_register_budget(isa):
    bad = isa_check.check_register_budget(isa)
    assert not bad, "; ".join(bad)


def test_no_compiler_vmem_between_asm_loads_and_their_waits(isa):
    bad = isa_check.check_vmem_windows(isa)
    assert not bad, "compiler-issued vector memory inside an asm load window: %r" % bad[:8]


def test_cell_shape(isa):
    """The single-query kernels pair a lane's two scores with v_pk_mad_i16 (no v_perm_b32 left); b128 profile reads."""
    bad = isa_check.check_cell_shape(isa)
    assert not bad, "; ".join(bad)


def test_stamp_ties_the_built_library_to_the_checked_sources(isa):
    """The stamp next to liboswald_hip.so names the library built from the sources that have just passed."""
    info = isa_check.stamp()
    on_disk = json.load(open(isa_check.STAMP))
    assert on_disk == info and info["library_sha256"] == isa_check.sha256_file(isa_check.LIB) and info["source_digest"] == isa_check.source_digest()
