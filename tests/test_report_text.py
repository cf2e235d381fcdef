"""the report `oswald -O search` prints, against the reference's own printf format strings.

Build-container only: the format strings are READ from /root/reference at test time (nothing of them is stored in
this repository) -- host/src/FPGAsearch.c:27-28, :60-65, :315-331 (header, query sections, footer) and
host/src/HybridSearch.c:620-622 (the hybrid mode's three calibration lines) -- turned into one regular expression
with the variable fields masked, and matched against the tool's stdout.  Where /root/reference does not exist (the
GPU box) the test is skipped.  The search runs in host-only mode (`-m 2`, no GPU needed): it prints through the same
print_header / print_report as `-m 0` and `-m 1`."""
import os
import re
import subprocess

import pytest

from oswald_amd import synth

import hostlib

REF = "/root/reference/host/src"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "FPGAsearch.c")), reason="the reference sources are not present here")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def printf_formats(fname, lo, hi):
    """Format strings of the printf calls on lines lo..hi (1-based, inclusive), C escapes resolved."""
    lines = open(os.path.join(REF, fname), errors="replace").read().split("\n")[lo - 1:hi]
    out = []
    for line in lines:
        for m in re.finditer(r'printf\s*\(\s*"((?:[^"\\]|\\.)*)"', line):
            out.append(m.group(1).encode().decode("unicode_escape"))
    return out


def format_regex(fmt):
    """A printf format as a regular expression: conversions become their value classes, the rest is literal.
    (An invalid conversion such as the reference's "% \\n" is printed by glibc as it stands.)"""
    rx, i = "", 0
    while i < len(fmt):
        c = fmt[i]
        if c != "%":
            rx += re.escape(c)
            i += 1
            continue
        m = re.match(r"%[-+0#]*\d*(?:\.\d+)?(ld|lf|lu|d|u|s|f|%)", fmt[i:])
        if not m:
            rx += re.escape("%")
            i += 1
            continue
        kind = m.group(1)
        rx += {"d": r"-?\d+", "ld": r"-?\d+", "u": r"\d+", "lu": r"\d+", "s": r"[^\n]*", "lf": r"-?\d+(?:\.\d+)?(?:e[-+]?\d+)?|inf|nan", "f": r"-?\d+(?:\.\d+)?",
               "%": "%"}[kind].join(("(?:", ")"))
        i += m.end()
    return rx


def test_report_matches_the_reference_format_strings(tmp_path):
    qs = synth.make_queries([40, 90, 130], seed=8)
    L, R, O = synth.make_database(120, qs, seed=9, homologs_per_query=2)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(120)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs, titles=["alpha query", "beta", "gamma | third"])
    hostlib.preprocess(str(tmp_path / "db.fasta"), str(tmp_path / "db"), 2)
    r = 7
    p = subprocess.run([hostlib.CLI, "-O", "search", "-m", "2", "-c", "2", "-r", str(r), "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    head = printf_formats("FPGAsearch.c", 27, 28) + printf_formats("FPGAsearch.c", 60, 65)
    query = printf_formats("FPGAsearch.c", 315, 318)
    hit = printf_formats("FPGAsearch.c", 319, 320)
    foot = printf_formats("FPGAsearch.c", 322, 331)
    assert len(head) == 8 and len(query) == 4 and len(hit) == 1 and len(foot) == 10, (head, query, hit, foot)
    assert hit[0] == "%d\t%s"     # the reference's titles keep their own newline; ours print it after the title
    rx = "".join(format_regex(f) for f in head)
    rx += ("".join(format_regex(f) for f in query) + (format_regex(hit[0]) + "\n") * r) * len(qs)
    # ctime() ends with its own newline: the date line has none in the format
    rx += format_regex(foot[0]) + "\n" + "".join(format_regex(f) for f in foot[1:])
    m = re.fullmatch(rx, p.stdout)
    if not m:   # say where the outputs part
        lines = p.stdout.split("\n")
        for k in range(len(lines), 0, -1):
            if re.match(rx[: len(rx)], "\n".join(lines[:k])):
                break
        pytest.fail("report differs from the reference's format strings; stdout was:\n" + p.stdout)
    # the fields the formats mask
    assert "Query description: \t\talpha query\n" in p.stdout or "Query description: \t\tbeta\n" in p.stdout
    assert f"Database size:\t\t\t120 sequences ({int(L.sum())} residues) \n" in p.stdout


def test_hybrid_calibration_lines_use_the_reference_formats():
    """`-m 1` needs a GPU; its three extra lines (reference HybridSearch.c:620-622) are checked in the tool's source:
    each reference format string must occur there verbatim (the reference's stray "% " conversion is spelt "%% ")."""
    src = open(os.path.join(ROOT, "oswald_amd", "host", "oswald_main.cpp")).read()
    fmts = printf_formats("HybridSearch.c", 620, 622)
    assert len(fmts) == 3
    for f in fmts:
        c_literal = f.replace("\\", "\\\\").replace("\t", "\\t").replace("\n", "\\n").replace("% \\n", "%% \\n")
        assert '"' + c_literal + '"' in src, c_literal
