#!/usr/bin/env python3
"""tools/isa_check.py -- ISA check of the hand-scheduled DP kernels and the stamp that ties the shipped library to it.

The column loop of osw_sw_pk16 / osw_sw_pk16q / osw_sw_s16 / osw_sw_s16q and (round 4) of the 8-bit kernel osw_sw_q8
loads straight into fixed physical registers from inline asm, two columns ahead (sw_kernels.hip, "Input registers of a
column step"; q8_cell.h for the 8-bit kernel's set).  The compiler does not know
about those loads, so the design rests on it never touching these registers itself, on there being no scratch spills
and on no compiler-issued vector-memory operation inside the loops that contain the asm loads (their waits are counted
by hand).  check() compiles the kernels to assembly with the hipcc at hand and verifies exactly that
(tests/test_isa_inflight.py runs it on CPUs).

The library that RUNS is the one built in this tree and shipped to the GPU box.  stamp() records, next to it, the
SHA-256 of the liboswald_hip.so whose sources passed the check, the digest of those sources and the compiler version;
tests/test_gpu_isa_guard.py (-m gpu) verifies on the GPU box that the library the process actually mapped is that file,
bit for bit -- a library rebuilt there by another compiler would not be."""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oswald_amd", "csrc")
SRC = os.path.join(CSRC, "sw_kernels.hip")
LIB = os.path.join(ROOT, "oswald_amd", "liboswald_hip.so")
STAMP = os.path.join(ROOT, "oswald_amd", "liboswald_hip.isa.json")
# kernel groups: their fixed in-flight registers (the #define that lists them, and the file it is in), the register
# budget of their launch bounds, and the scratch bytes per lane tolerated OUTSIDE the column loops (loop-invariant state
# parked before a round; inside the loops check_vmem_windows allows none)
GROUPS = (
    dict(kernels=("osw_sw_pk16", "osw_sw_pk16q", "osw_sw_pk16qt", "osw_sw_s16", "osw_sw_s16q", "osw_sw_s16qt"), define="OSW_INFLIGHT", file="sw_kernels.hip",
         budget=168,   # three waves per SIMD; the compiler gets 140, the asm statements 27 fixed ones (v158 is spare; v148, v149 unused)
         scratch=64, min_asm_uses=1000, nfixed=27),   # (spills of item-level values around the rounds: never inside a column loop, check_vmem_windows;
                                                     # round 6: 52 B in osw_sw_s16qt, which holds the single-query cell for its tails beside the pair cell)
    dict(kernels=("osw_sw_i32",), define="OSW_INFLIGHT", file="sw_kernels.hip",
         budget=168,   # the hand-scheduled int32 cell (cell_bits = 32): the int16 kernels' column loop and register budget
         scratch=0, min_asm_uses=500, nfixed=27),
    dict(kernels=("osw_sw_i32r",), define="OSW_INFLIGHT", file="sw_kernels.hip",
         budget=168,   # the same cell in the re-run pipeline (4-row strips only)
         scratch=0, min_asm_uses=40, nfixed=27,
         windows=False),  # (its batch loop polls / publishes progress words between the steps -- LDS accesses -- and stores the item's score behind
                          # the rounds; the linear scan of check_vmem_windows does not follow that control flow)
    dict(kernels=("osw_sw_q8",), define="OSW8_INFLIGHT", file="q8_cell.h",
         budget=80,    # six waves per SIMD; the compiler gets 72, the asm 8 more
         scratch=0, min_asm_uses=100, nfixed=8),
)
KERNELS = tuple(k for g in GROUPS for k in g["kernels"])
SOURCES = ("sw_kernels.hip", "sw_kernels.h", "q8_cell.h", "oswald_hip.cpp", "osw_planner.inc")


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return p if os.path.exists(p) else None


def reserved_registers(group=GROUPS[0]):
    m = re.search(r'#define %s (.*)' % group["define"], open(os.path.join(CSRC, group["file"])).read())
    regs = [int(x) for x in re.findall(r'"v(\d+)"', m.group(1))]
    assert len(regs) == group["nfixed"], (group["define"], regs)
    return set(regs)


def compile_to_asm():
    """The kernels' assembly, compiled by the library's own Makefile (`make asm`: same compiler, same flags as the
    shipped liboswald_hip.so, so that what is checked is what is shipped)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sw_kernels.s")
        subprocess.check_call(["make", "-s", "-C", CSRC, "asm", "ASM_OUT=" + out, "HIPCC=" + hipcc_path()], stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def _touches(code, reserved):
    for m in re.finditer(r'\bv(\d+)\b', code):
        if int(m.group(1)) in reserved:
            return True
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', code):
        if any(r in reserved for r in range(int(m.group(1)), int(m.group(2)) + 1)):
            return True
    return False


def check_inflight_registers(isa, group=GROUPS[0]):
    """-> (instructions of the asm blocks that use the fixed registers, [(line, code)] of compiler-scheduled ones that do)"""
    reserved = reserved_registers(group)
    KERNELS = group["kernels"]
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
    return seen_asm_use, bad


def check_register_budget(isa, group=GROUPS[0]):
    """The kernels' launch bounds ask for group["budget"] registers at most (waves per SIMD = 512 / budget); a spill of
    loop-invariant values outside the column loops is tolerated (group["scratch"] bytes), spill traffic inside them is
    ruled out by check_vmem_windows.  -> list of complaints"""
    text, bad = "\n".join(isa), []
    VGPR_BUDGET = group["budget"]
    for k in group["kernels"]:
        m = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)' % k, text)
        m2 = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)' % k, text)
        if not m or int(m.group(1)) > VGPR_BUDGET:
            bad.append("%s needs more than %d VGPRs" % (k, VGPR_BUDGET))
        if not m2 or int(m2.group(1)) > group["scratch"]:
            bad.append("%s spills %s bytes per lane" % (k, m2.group(1) if m2 else "?"))
    return bad


def check_vmem_windows(isa, group=GROUPS[0]):
    """Inside the column loops every vector-memory instruction must come from the asm blocks: a compiler-issued one would shift
    the hand-counted waits.  A column loop, in the text of a kernel: from the first step's opening wait (the asm
    `s_waitcnt vmcnt(5) lgkmcnt(0)`) to the last asm vector-memory instruction in front of the round's closing wait (the asm
    `s_waitcnt vmcnt(0) lgkmcnt(0)`).  A spill of a loop-invariant value between a round's prologue -- which ends in
    `s_waitcnt vmcnt(0)` -- and its first step, or its reload behind the last step, is outside (counted by check_register_budget)."""
    KERNELS = group["kernels"]
    fn, inasm, bad = None, False, []
    first_begin, last_asm_vm, pending = None, None, []   # of the loop being scanned; compiler VM ops seen since first_begin
    for i, line in enumerate(isa):
        m = re.match(r'^(osw_\w+):', line)
        if m:
            fn, first_begin, last_asm_vm, pending = m.group(1), None, None, []
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        if fn not in KERNELS or not line.startswith("\t"):
            continue
        code = line.split(";")[0].strip()
        vm = re.match(r'(global_|buffer_|flat_|scratch_)', code) is not None
        if inasm and code.startswith("s_waitcnt vmcnt(5) lgkmcnt(0)") and first_begin is None:
            first_begin, pending = i, []
        elif inasm and code.startswith("s_waitcnt vmcnt(0) lgkmcnt(0)"):
            if first_begin is not None and last_asm_vm is not None:
                bad += [(k + 1, c) for k, c in pending if k < last_asm_vm]
            first_begin, last_asm_vm, pending = None, None, []
        elif inasm and vm and first_begin is not None:
            last_asm_vm = i
        elif not inasm and vm and first_begin is not None:
            pending.append((i, code))
    return bad


def check_cell_shape(isa):
    """The rows of the int16 cells as they are priced: the single-query kernels pair the scores of a lane's two
    sequences with one v_pk_mad_i16 per row and have no v_perm_b32 left; every int16 kernel reads its profile with
    ds_read_b128 only and keeps v_pk_maximum3_f16 as its maximum.  -> list of complaints"""
    counts, fn = {}, None
    for line in isa:
        m = re.match(r'^(osw_\w+):', line)
        if m:
            fn = m.group(1)
        if fn not in GROUPS[0]["kernels"] or not line.startswith("\t"):
            continue
        op = line.split(";")[0].strip().split(" ")[0]
        counts.setdefault(fn, {}).setdefault(op, 0)
        counts[fn][op] += 1
    bad = []
    for k in GROUPS[0]["kernels"]:
        c = counts.get(k, {})
        single = not k.endswith(("q", "qt"))
        tails = k.endswith("qt")
        if c.get("v_perm_b32", 0):
            bad.append("%s has %d v_perm_b32" % (k, c["v_perm_b32"]))
        if single and c.get("v_pk_mad_i16", 0) < 300:
            bad.append("%s has %d v_pk_mad_i16 (one per row of every unrolled strip height expected)" % (k, c.get("v_pk_mad_i16", 0)))
        if not single and not tails and c.get("v_pk_mad_i16", 0):
            bad.append("%s has %d v_pk_mad_i16 (the query-pair row adds with a 32-bit add)" % (k, c["v_pk_mad_i16"]))
        if tails:
            # the query-pair ROW adds with a 32-bit add; the v_pk_mad_i16 of a ...qt kernel are the rows of the single-query cell that runs the
            # TAILS of its SHORT items (OswSearchArgs::hand): whole unrolled strip sets of 312 rows (4 + 8 + .. + 48), and beside them at least
            # the pair instantiations (cells x ordinary / SHORT x two passes) at 3.5 maxima per row
            mad, mx = c.get("v_pk_mad_i16", 0), c.get("v_pk_maximum3_f16", 0)
            if mad == 0 or mad % 312 or mad > 4 * 312 or mx - 3.5 * mad < 3.5 * 312 * (8 if k == "osw_sw_s16qt" else 4):  # (osw_sw_pk16qt: one cell)
                bad.append("%s has %d v_pk_mad_i16 and %d v_pk_maximum3_f16 (tail rows come in sets of 312; the pair rows add with a 32-bit add)" % (k, mad, mx))
        if c.get("ds_read_b64", 0) or c.get("ds_read2_b64", 0) or c.get("ds_read_b128", 0) < 100:
            bad.append("%s reads its profile with %d ds_read_b128, %d ds_read_b64" % (k, c.get("ds_read_b128", 0), c.get("ds_read_b64", 0)))
        if c.get("v_pk_maximum3_f16", 0) < 1000:
            bad.append("%s has %d v_pk_maximum3_f16" % (k, c.get("v_pk_maximum3_f16", 0)))
    return bad


def check_int32_cell(isa):
    """The exact kernels, hand-scheduled since the second session of round 5: osw_sw_i32 (whole searches with cell_bits = 32, strips of
    4 .. 48 rows) and osw_sw_i32r (the re-run pipeline, 4-row strips).  The row is v_add_u32_sdwa + 3.5 v_max3_i32 + 2 v_subrev_u32 -- no
    unpacking of the int16 scores (v_bfe_i32 / v_ashrrev_i32), no three-operand add --, profile by ds_read_b64, <= 168 VGPRs, no
    scratch at all.  -> list of complaints"""
    text, bad, counts, fn = "\n".join(isa), [], {}, None
    for line in isa:
        m = re.match(r'^(osw_\w+):', line)
        if m:
            fn = m.group(1)
        if fn in ("osw_sw_i32", "osw_sw_i32r") and line.startswith("\t"):
            op = line.split(";")[0].strip().split(" ")[0]
            counts.setdefault(fn, {}).setdefault(op, 0)
            counts[fn][op] += 1
    for k, min_rows in (("osw_sw_i32", 600), ("osw_sw_i32r", 8)):
        m = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)' % k, text)
        m2 = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)' % k, text)
        if not m or int(m.group(1)) > 168:
            bad.append("%s needs %s VGPRs" % (k, m.group(1) if m else "?"))
        if not m2 or int(m2.group(1)) != 0:
            bad.append("%s spills %s bytes per lane" % (k, m2.group(1) if m2 else "?"))
        c = counts.get(k, {})
        if c.get("scratch_load_dword", 0) or c.get("scratch_store_dword", 0):
            bad.append("%s has scratch traffic" % k)
        rows = c.get("v_add_u32_sdwa", 0)
        if rows < min_rows or c.get("v_max3_i32", 0) < 3.4 * rows or c.get("v_bfe_i32", 0) or c.get("v_ashrrev_i32_e32", 0) or c.get("v_add3_u32", 0) > 60:
            bad.append("%s: %d v_add_u32_sdwa, %d v_max3_i32, %d v_bfe_i32, %d v_add3_u32 (one SDWA add and 3.5 maxima per row, no unpacking)"
                       % (k, rows, c.get("v_max3_i32", 0), c.get("v_bfe_i32", 0), c.get("v_add3_u32", 0)))
        if c.get("ds_read_b64", 0) < min_rows // 4 or c.get("ds_read_b128", 0):
            bad.append("%s reads its profile with %d ds_read_b64, %d ds_read_b128" % (k, c.get("ds_read_b64", 0), c.get("ds_read_b128", 0)))
    return bad


def check(isa=None):
    """All checks; raises AssertionError with the findings."""
    isa = isa or compile_to_asm()
    shape = check_cell_shape(isa)
    assert not shape, "; ".join(shape)
    i32 = check_int32_cell(isa)
    assert not i32, "; ".join(i32)
    for group in GROUPS:
        seen, bad = check_inflight_registers(isa, group)
        assert seen > group["min_asm_uses"], "%s: the asm blocks that use the fixed registers were not found" % (group["kernels"],)
        assert not bad, "compiler-scheduled instructions touch in-flight registers: %r" % bad[:8]
        budget = check_register_budget(isa, group)
        assert not budget, "; ".join(budget)
        vm = check_vmem_windows(isa, group) if group.get("windows", True) else []
        assert not vm, "compiler-issued vector memory inside an asm load window: %r" % vm[:8]
    return isa


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def source_digest():
    h = hashlib.sha256()
    for rel in SOURCES:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stamp():
    """Check the ISA of the current sources, make sure liboswald_hip.so is built from them (make is a no-op when it is
    up to date) and write the stamp."""
    check()
    subprocess.check_call(["make", "-s", "-C", CSRC])
    ver = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    info = {"library_sha256": sha256_file(LIB), "source_digest": source_digest(), "hipcc": ver[0] if ver else "", "kernels": list(KERNELS),
            "checks": ["in-flight registers untouched by compiler-scheduled code",
                       "; ".join("%s: <= %d VGPRs, <= %d B of scratch outside the column loops" % ("/".join(g["kernels"]), g["budget"], g["scratch"]) for g in GROUPS),
                       "no compiler-issued vector memory (spill traffic included) inside the asm load windows",
                       "single-query int16 kernels: one v_pk_mad_i16 per row, no v_perm_b32; profile reads are ds_read_b128",
                       "osw_sw_i32 / osw_sw_i32r: hand-scheduled int32 rows (v_add_u32_sdwa + 3.5 v_max3_i32, ds_read_b64), <= 168 VGPRs, no scratch"]}
    with open(STAMP, "w") as f:
        json.dump(info, f, indent=1)
    return info


if __name__ == "__main__":
    if not hipcc_path():
        sys.exit("hipcc not available")
    print(json.dumps(stamp(), indent=1))
