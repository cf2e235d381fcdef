"""ISA check of the packed-int16 kernels (CPU test: hipcc cross-compiles without a GPU).

The column loop of osw_sw_pk16 / osw_sw_pk16q loads straight into fixed physical
registers from inline asm, two columns ahead (sw_kernels.hip, "Input registers of
a column step").  The compiler does not know about those loads, so the design
rests on it never touching these registers itself.  This test compiles the
kernels to assembly and checks exactly that, plus the things the hand-counted
waits rely on: no scratch spills and no compiler-issued vector-memory operation
inside the loops that contain the asm loads.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oswald_amd", "csrc", "sw_kernels.hip")
KERNELS = ("osw_sw_pk16", "osw_sw_pk16q", "osw_sw_s16", "osw_sw_s16q")


def _reserved():
    text = open(SRC).read()
    m = re.search(r'#define OSW_INFLIGHT (.*)', text)
    regs = [int(x) for x in re.findall(r'"v(\d+)"', m.group(1))]
    assert len(regs) == 8
    return set(regs)


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "sw_kernels.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(SRC), "-o", str(out), SRC], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def _touches(code, reserved):
    for m in re.finditer(r'\bv(\d+)\b', code):
        if int(m.group(1)) in reserved:
            return True
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', code):
        if any(r in reserved for r in range(int(m.group(1)), int(m.group(2)) + 1)):
            return True
    return False


def test_compiler_never_touches_the_inflight_registers(isa):
    reserved = _reserved()
    fn, inasm, bad, seen_asm_use = None, False, [], 0
    for i, line in enumerate(isa):
        m = re.match(r'^(osw_\w+):', line)
        if m:
            fn = m.group(1)
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        if fn not in KERNELS or not line.startswith("\t"):
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        if _touches(code, reserved):
            if inasm:
                seen_asm_use += 1
            else:
                bad.append((i + 1, code))
    assert seen_asm_use > 1000, "the asm blocks that use the fixed registers were not found"
    assert not bad, "compiler-scheduled instructions touch in-flight registers: %r" % bad[:8]


def test_register_budget(isa):
    """4 waves per SIMD need <= 128 VGPRs; a spill of a loop-invariant value outside the column loops is
    tolerated (a few bytes), spill traffic inside them is ruled out by the window test below."""
    text = "\n".join(isa)
    for k in KERNELS:
        m = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)' % k, text)
        m2 = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)' % k, text)
        assert m and int(m.group(1)) <= 128, "%s needs more than 128 VGPRs (4 waves per SIMD)" % k
        assert m2 and int(m2.group(1)) <= 32, "%s spills %s bytes per lane" % (k, m2.group(1))


def test_no_compiler_vmem_between_asm_loads_and_their_waits(isa):
    """Inside the column loops (the code between the prologue's asm loads and the
    final `s_waitcnt vmcnt(0)` of a round) every vector-memory instruction must
    come from the asm blocks: a compiler-issued one would shift the counts."""
    fn, inasm, window, bad = None, False, False, []
    for i, line in enumerate(isa):
        m = re.match(r'^(osw_\w+):', line)
        if m:
            fn, window = m.group(1), False
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        if fn not in KERNELS or not line.startswith("\t"):
            continue
        code = line.split(";")[0].strip()
        if inasm and code.startswith("global_load_ushort"):
            window = True
        if inasm and code.startswith("s_waitcnt vmcnt(0) lgkmcnt(0)"):
            window = False
        if window and not inasm and re.match(r'(global_|buffer_|flat_|scratch_)', code):
            bad.append((i + 1, code))
    assert not bad, "compiler-issued vector memory inside an asm load window: %r" % bad[:8]
