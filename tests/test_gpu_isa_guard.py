"""GPU box: the liboswald_hip.so this process actually mapped is, bit for bit, the one whose kernels passed the ISA
check in the build container (tools/isa_check.py: the hand-scheduled column loops keep loads in flight in fixed
physical registers; a library rebuilt by another compiler could break that silently, and only wrong scores would tell).
The stamp is written by `__graft_entry__.build()` / tests/test_isa_inflight.py and travels with the library."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loaded_library_is_the_isa_checked_one(hip_ctx):
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_check.py"))
    isa_check = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa_check)
    mapped = sorted({line.split()[-1] for line in open("/proc/self/maps") if "liboswald_hip" in line and line.rstrip().endswith(".so")})
    assert len(mapped) == 1, mapped
    assert os.path.realpath(mapped[0]) == os.path.realpath(isa_check.LIB), "the library was not loaded from the tree"
    assert os.path.exists(isa_check.STAMP), "no ISA stamp next to the library: run `python tools/isa_check.py` (or __graft_entry__.build()) where it was built"
    stamp = json.load(open(isa_check.STAMP))
    assert stamp["library_sha256"] == isa_check.sha256_file(mapped[0]), "the loaded library is not the one whose ISA was checked"
    assert stamp["source_digest"] == isa_check.source_digest(), "the kernel sources changed after the library was checked"
    if isa_check.hipcc_path():   # the box has a compiler too: its ISA of the same sources must pass as well
        isa_check.check()
