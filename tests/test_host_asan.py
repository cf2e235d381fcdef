"""The host layer under AddressSanitizer + UBSan (CPU only).  `make -C oswald_amd/host asan` builds the command line
tool with the host sources compiled in under -fsanitize=address,undefined; this file drives it through
`-O preprocess` and `-O search -m 2` (neither needs a GPU) with well-formed inputs and with the malformed ones a
parser of untrusted files must survive: FASTA without a trailing newline or '>' line, truncated <db>.seq, <db>.info
whose counts are larger than the file (or negative, or not numbers), a length table that is not sorted or does not add
up, <db>.g16 with a lying header, a missing <db>.desc.  Every run must end with the tool's own exit code and message
(the reference prints and exits with 2 / 3 on file errors, host/src/sequences.c:18, :133, :1110) -- never with a
sanitizer report."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from oswald_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_CLI = os.path.join(ROOT, "oswald_amd", "oswald_asan")


@pytest.fixture(scope="module")
def asan_cli():
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oswald_amd", "host"), "asan"], capture_output=True, text=True)
    if p.returncode != 0 or not os.path.exists(ASAN_CLI):
        pytest.skip("no sanitizer runtime for this compiler: " + p.stderr[-300:])
    return ASAN_CLI


def run(cli, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98",
               OMP_NUM_THREADS="2")
    p = subprocess.run([cli] + list(args), capture_output=True, text=True, env=env, timeout=300)
    assert "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr and p.returncode not in (98, 99), p.stderr[-3000:]
    assert p.returncode >= 0, f"killed by signal {-p.returncode}: {p.stderr[-1000:]}"
    return p


def make_db(tmp_path, cli, nseq=200, seed=3):
    qs = synth.make_queries([40, 130], seed=seed)
    L, R, O = synth.make_database(nseq, qs, seed=seed + 1, homologs_per_query=2)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(nseq)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs, titles=["first query", "second query"])
    p = run(cli, "-O", "preprocess", "-i", str(tmp_path / "db.fasta"), "-o", str(tmp_path / "db"))
    assert p.returncode == 0, p.stderr
    return str(tmp_path / "db"), str(tmp_path / "q.fasta"), L


def search(cli, db, q, *more):
    return run(cli, "-O", "search", "-m", "2", "-c", "2", "-q", q, "-d", db, *more)


def test_well_formed_runs_clean(tmp_path, asan_cli):
    db, q, L = make_db(tmp_path, asan_cli)
    p = search(asan_cli, db, q)
    assert p.returncode == 0 and "Query no.\t\t\t2" in p.stdout and f"({int(L.sum())} residues)" in p.stdout
    first = p.stdout
    os.environ["OSWALD_NO_GROUP_CACHE"] = "1"
    try:
        p = search(asan_cli, db, q, "-k", "60000", "-r", "500")      # several chunks, r larger than the database, no cache
    finally:
        del os.environ["OSWALD_NO_GROUP_CACHE"]
    assert p.returncode == 0 and p.stdout.count("\nQuery no.") == 2
    assert [l for l in first.split("\n") if l[:1].isdigit()][:3] == [l for l in p.stdout.split("\n") if l[:1].isdigit()][:3]


def test_fasta_corner_cases(tmp_path, asan_cli):
    cases = {
        "no_trailing_newline": ">a\nACDEFGHIKL\n>b\nMNPQRSTVWY",          # the reference loses the last residue here (and segfaults on some inputs)
        "crlf_and_blank_lines": ">a\r\nACDEF\r\n\r\n>b\r\nGHIK\r\n",
        "lowercase_and_odd_bytes": ">a\nacdefJOUBZX*-\n>b\n\x01\x7f\xff\n",
        "empty_record": ">a\n>b\nACD\n",
        "only_titles": ">a\n>b\n",
        "long_line": ">a\n" + "ACDEFGHIKLMNPQRSTVWY" * 3000 + "\n",
    }
    for name, text in cases.items():
        f = tmp_path / f"{name}.fasta"
        f.write_bytes(text.encode("latin-1"))
        p = run(asan_cli, "-O", "preprocess", "-i", str(f), "-o", str(tmp_path / name))
        assert p.returncode == 0, (name, p.stderr)
        q = search(asan_cli, str(tmp_path / name), str(f))
        assert q.returncode == 0, (name, q.stdout[-300:], q.stderr[-300:])
    for name, text in {"no_title": "ACDEF\n>b\nACD\n", "too_long": ">a\n" + "A" * 70000 + "\n"}.items():
        f = tmp_path / f"{name}.fasta"
        f.write_text(text)
        p = run(asan_cli, "-O", "preprocess", "-i", str(f), "-o", str(tmp_path / name))
        assert p.returncode == 2 and "OSWALD" in p.stdout
    (tmp_path / "empty.fasta").write_text("")
    p = run(asan_cli, "-O", "preprocess", "-i", str(tmp_path / "empty.fasta"), "-o", str(tmp_path / "empty"))
    assert p.returncode == 0
    p = run(asan_cli, "-O", "preprocess", "-i", str(tmp_path / "does_not_exist.fasta"), "-o", str(tmp_path / "x"))
    assert p.returncode == 2 and "opening input sequence file" in p.stdout


def test_malformed_database_files(tmp_path, asan_cli):
    db, q, L = make_db(tmp_path, asan_cli)
    good = {ext: open(db + ext, "rb").read() for ext in (".info", ".seq", ".desc", ".g16")}
    nseq, D = len(L), int(L.sum())

    def restore():
        for ext, data in good.items():
            with open(db + ext, "wb") as f:
                f.write(data)

    def expect_error(what, text=None):
        p = search(asan_cli, db, q)
        assert p.returncode == 2, (what, p.returncode, p.stdout[-300:], p.stderr[-300:])
        assert "OSWALD" in p.stdout and (text is None or text in p.stdout), (what, p.stdout[-300:])
        restore()

    # <db>.seq cut short: in the length table, in the residues, by one byte
    for cut in (nseq, 2 * nseq + D // 2, 2 * nseq + D - 1):
        with open(db + ".seq", "r+b") as f:
            f.truncate(cut)
        expect_error(f"seq truncated to {cut}", "shorter than its info file says")
    # ... or longer than the info file says
    with open(db + ".seq", "ab") as f:
        f.write(b"\x00" * 10)
    expect_error("seq too long", "does not have the size")
    # <db>.info with counts larger than the file, absurd, negative, or not numbers
    for text in (f"{nseq * 1000} {D} 40", f"{nseq} {D * 1000} 40", "99999999999999999 99999999999999999 40", f"-5 {D} 40", f"{nseq} -7 40", "abc def", ""):
        with open(db + ".info", "w") as f:
            f.write(text)
        expect_error(f"info '{text}'")
    # a length table that does not add up / is not sorted (the group layout relies on both)
    lens = np.frombuffer(good[".seq"][:2 * nseq], dtype=np.uint16).copy()
    bad = lens.copy(); bad[-1] += 9
    with open(db + ".seq", "r+b") as f:
        f.write(bad.tobytes())
    expect_error("lengths do not add up", "do not add up")
    bad = lens.copy(); bad[3], bad[-3] = bad[-3], bad[3]
    with open(db + ".seq", "r+b") as f:
        f.write(bad.tobytes())
    expect_error("lengths not sorted", "not sorted")
    # <db>.g16 with a lying header (more groups / more bytes than the file holds), garbage, truncated, empty: ignored with a warning
    hdr = bytearray(good[".g16"][:80])
    for off, val in ((32, 10**12), (40, 10**12), (16, 10**9), (24, 1)):           # groups, vD, sequences_count, D
        h2 = bytearray(hdr); struct.pack_into("<Q", h2, off, val)
        with open(db + ".g16", "wb") as f:
            f.write(bytes(h2) + good[".g16"][80:])
        p = search(asan_cli, db, q)
        assert p.returncode == 0 and "does not match the database" in p.stderr, (off, p.stderr[-300:])
        restore()
    for blob in (b"", b"OSWG16\0\0", os.urandom(5000), good[".g16"][:len(good[".g16"]) // 2]):
        with open(db + ".g16", "wb") as f:
            f.write(blob)
        p = search(asan_cli, db, q)
        assert p.returncode == 0, p.stderr[-300:]
        restore()
    # residues changed under the cache (same size, same lengths): the cache must not be used
    with open(db + ".seq", "r+b") as f:
        f.seek(2 * nseq + D // 3)
        f.write(bytes([7]))
    p = search(asan_cli, db, q)
    assert p.returncode == 0 and "does not match the database" in p.stderr
    restore()
    # a missing / short description file: the reference exits(3) on the former; short files give empty titles
    os.remove(db + ".desc")
    expect_error("no desc", "sequence description file")
    with open(db + ".desc", "wb") as f:
        f.write(good[".desc"][:50])
    p = search(asan_cli, db, q)
    assert p.returncode == 0
    restore()
    # queries: missing file, empty file
    p = search(asan_cli, db, str(tmp_path / "nope.fasta"))
    assert p.returncode == 2
    (tmp_path / "noq.fasta").write_text("")
    p = search(asan_cli, db, str(tmp_path / "noq.fasta"))
    assert p.returncode in (0, 2)
