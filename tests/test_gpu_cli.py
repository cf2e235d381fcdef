"""End to end through the command line tool on a GPU: `oswald -O preprocess`
then `oswald -O search -m 0`, report parsed and compared with the oracle's
top-r (scores, titles, tie order)."""
import os
import re
import subprocess

import numpy as np
import pytest

from oswald_amd import dblayout, submat, synth

import hostlib
from helpers import pack_queries

pytestmark = pytest.mark.gpu


def parse_report(text):
    blocks = []
    cur = None
    for line in text.split("\n"):
        if line.startswith("Query no."):
            cur = {"hits": []}
            blocks.append(cur)
        elif cur is not None and line.startswith("Query description:"):
            cur["title"] = line.split("\t")[-1]
        elif cur is not None and line.startswith("Query length:"):
            cur["m"] = int(re.search(r"(\d+) residues", line).group(1))
        elif cur is not None and re.match(r"^\d+\t", line):
            s, t = line.split("\t", 1)
            cur["hits"].append((int(s), t))
        elif line.startswith("Search date"):
            cur = None
    return blocks


@pytest.mark.parametrize("matrix,go,ge,extra", [("blosum62", 10, 2, []), ("pam250", 14, 2, ["-k", "60000"]),
                                                ("blosum62", 10, 2, ["-k", "60000", "-f", "2"])])   # two context devices: async uploads per round
def test_cli_search_report(tmp_path, oracle, matrix, go, ge, extra):
    qs = synth.make_queries([120, 45, 300], seed=3)
    L, R, O = synth.make_database(600, qs, seed=9, homologs_per_query=4)
    seqs = [R[O[i]:O[i + 1]] for i in range(600)]
    titles = [f"syn|{i}|len={len(s)}" for i, s in enumerate(seqs)]
    synth.write_fasta(str(tmp_path / "db.fasta"), seqs, titles)
    synth.write_fasta(str(tmp_path / "q.fasta"), qs, titles=["q120 first in file", "q45", "q300"])
    db = str(tmp_path / "db")
    subprocess.run([hostlib.CLI, "-O", "preprocess", "-i", str(tmp_path / "db.fasta"), "-o", db], check=True, capture_output=True)
    p = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-q", str(tmp_path / "q.fasta"), "-d", db, "-s", matrix,
                        "-g", str(go), "-e", str(ge), "-r", "7"] + extra, capture_output=True, text=True,
                       env=dict(os.environ, OSWALD_DEVICE_IDS="0,0"))   # (used only with -f 2: both on the box's one GPU)
    assert p.returncode == 0, p.stderr
    out = p.stdout
    assert out.startswith("\nOSWALD v1.0 \n\nDatabase file:\t\t\t" + db + "\n")
    assert f"Database size:\t\t\t600 sequences ({int(L.sum())} residues) \n" in out
    assert f"Substitution matrix:\t\t{matrix.upper()}\nGap open penalty:\t\t{go}\nGap extend penalty:\t\t{ge}\n" in out
    assert re.search(r"Search speed:\t\t\t\d+\.\d\d GCUPS\n", out)
    blocks = parse_report(out)
    # queries are reported in ascending length order (reference sequences.c:342)
    assert [b["m"] for b in blocks] == [45, 120, 300]
    assert [b["title"] for b in blocks] == ["q45", "q120 first in file", "q300"]
    # expected: oracle scores on the sorted database, reference tie order
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    sorted_qs = [qs[1], qs[0], qs[2]]
    a, m, ad = pack_queries(sorted_qs)
    want = oracle.search_chunk_scalar(a, m, ad, b, n, disp.astype(np.uint32), 16, submat.load(matrix), go, ge)
    for qi, blk in enumerate(blocks):
        sc, ix = dblayout.topr_reference_order(want[qi, :600], 7)
        assert [h[0] for h in blk["hits"]] == sc.tolist()
        assert [h[1] for h in blk["hits"]] == [titles[order[i]] for i in ix]


def test_cli_info_and_footer(tmp_path):
    """`-O info` lists the device; the search footer keeps the reference's labels
    (reference FPGAsearch.c:322-331) so that scripts parsing OSWALD reports keep working."""
    p = subprocess.run([hostlib.CLI, "-O", "info"], capture_output=True, text=True)
    assert p.returncode == 0 and "Device 0:" in p.stdout and "compute units:" in p.stdout and "gfx950" in p.stdout
    qs = synth.make_queries([33], seed=1)
    L, R, O = synth.make_database(64, qs, seed=2, homologs_per_query=1)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(64)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs, titles=["the query"])
    db = str(tmp_path / "db")
    subprocess.run([hostlib.CLI, "-O", "preprocess", "-i", str(tmp_path / "db.fasta"), "-o", db], check=True, capture_output=True)
    s = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-q", str(tmp_path / "q.fasta"), "-d", db, "-c", "7", "-b", "128", "-r", "100"],
                       capture_output=True, text=True)
    assert s.returncode == 0, s.stderr
    lines = s.stdout.split("\n")
    i = next(k for k, l in enumerate(lines) if l.startswith("Search date:"))
    assert lines[i].startswith("Search date:\t\t\t")
    assert re.fullmatch(r"Search time:\t\t\t\d+\.\d{6} seconds", lines[i + 1])
    assert re.fullmatch(r"Search speed:\t\t\t\d+\.\d\d GCUPS", lines[i + 2])
    assert lines[i + 10] == ""      # nothing after the reference's last footer line
    assert lines[i + 3:i + 10] == ["CPU threads:\t\t\t7", "CPU vector length:\t\t16", "CPU block width:\t\t128", "Number of FPGAs:\t\t1",
                                   "FPGA vector length:\t\t16", "FPGA block width:\t\t28", "Max. chunk size in FPGA:\t134217728 bytes"]
    # -r larger than the database is clipped to the database size (reference FPGAsearch.c:68)
    assert sum(1 for l in lines if re.match(r"^\d+\tsyn\|", l)) == 64


@pytest.mark.parametrize("ndev", [1, 2])
def test_cli_hybrid_mode_equals_accelerator_mode(tmp_path, ndev):
    """`-m 1` (the reference's default): a test portion of the database on both sides, the rest divided in the
    measured proportion, host and GPU at the same time (HybridSearch.c:124-228, :620-631).  Same query sections as
    `-m 0`, plus the three calibration lines of the reference's hybrid report."""
    qs = synth.make_queries([150, 61, 300], seed=41)
    L, R, O = synth.make_database(2000, qs, seed=43, homologs_per_query=3)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(2000)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs)
    db = str(tmp_path / "db")
    subprocess.run([hostlib.CLI, "-O", "preprocess", "-i", str(tmp_path / "db.fasta"), "-o", db], check=True, capture_output=True)
    common = ["-q", str(tmp_path / "q.fasta"), "-d", db, "-r", "15", "-k", "300000"]
    gpu = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0"] + common, capture_output=True, text=True)
    assert gpu.returncode == 0, gpu.stderr
    # (run ndev context devices on the one GPU of the box)
    hyb = subprocess.run([hostlib.CLI, "-O", "search", "-m", "1", "-p", "0.05", "-c", "8", "-f", str(ndev)] + common, capture_output=True, text=True,
                         env=dict(os.environ, OSWALD_DEVICE_IDS=",".join(["0"] * ndev)))
    assert hyb.returncode == 0, hyb.stderr
    assert parse_report(hyb.stdout) == parse_report(gpu.stdout)
    assert re.search(r"Test DB percentage:\t\t0\.0500% \nCPU estimated speed:\t\t\d+\.\d\d GCUPS\nFPGA estimated speed:\t\t\d+\.\d\d GCUPS\n", hyb.stdout)
    host = subprocess.run([hostlib.CLI, "-O", "search", "-m", "2", "-c", "8"] + common, capture_output=True, text=True)
    assert host.returncode == 0 and parse_report(host.stdout) == parse_report(gpu.stdout)
    # A database this small is one piece for the accelerator after the test.  With the test hook both sides take work from their
    # ends until they meet (the accelerator in several pieces, the host in batches), the host's candidates are merged with the
    # devices' top lists -- and the report is the same, ties included.
    split = subprocess.run([hostlib.CLI, "-O", "search", "-m", "1", "-p", "0.05", "-c", "8", "-f", str(ndev)] + common, capture_output=True, text=True,
                           env=dict(os.environ, OSWALD_DEVICE_IDS=",".join(["0"] * ndev), OSWALD_HYBRID_TEST_SPLIT="1", OSWALD_DEBUG_PHASES="1"))
    assert split.returncode == 0, split.stderr
    assert parse_report(split.stdout) == parse_report(gpu.stdout)
    m = re.search(r"host done at .*: groups (\d+) \.\. (\d+) of", split.stderr)
    assert m and int(m.group(1)) < int(m.group(2)), split.stderr[-600:]          # the host did take part of the database
    # -r beyond what the devices select: the reference's static division with the score table on the host
    long_gpu = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", "1500"] + common[:-4] + ["-k", "300000"], capture_output=True, text=True)
    long_hyb = subprocess.run([hostlib.CLI, "-O", "search", "-m", "1", "-p", "0.05", "-c", "8", "-r", "1500"] + common[:-4] + ["-k", "300000"], capture_output=True, text=True)
    assert long_gpu.returncode == 0 and long_hyb.returncode == 0, long_hyb.stderr
    assert parse_report(long_hyb.stdout) == parse_report(long_gpu.stdout)


def _db_and_queries(tmp_path, nseq, qlens, seed):
    qs = synth.make_queries(qlens, seed=seed)
    L, R, O = synth.make_database(nseq, qs, seed=seed + 2, homologs_per_query=3)
    synth.write_fasta(str(tmp_path / "db.fasta"), [R[O[i]:O[i + 1]] for i in range(nseq)])
    synth.write_fasta(str(tmp_path / "q.fasta"), qs)
    db = str(tmp_path / "db")
    subprocess.run([hostlib.CLI, "-O", "preprocess", "-i", str(tmp_path / "db.fasta"), "-o", db], check=True, capture_output=True)
    return db


@pytest.mark.parametrize("ids", ["0,0", "0,0,0", "0,1", "0,1,0"])
def test_cli_several_devices_deal_the_database(tmp_path, ids):
    """`oswald -m 0 -f N`: the length-sorted database is dealt to the N devices in blocks of 128 sequences (index maps
    through oswald_hip_chunk_set_index), several pieces per device at this -k, uploads of the next round queued while
    the current one is searched, lists gathered on the GPUs (RCCL between distinct GPUs) -- and the report is the one
    `-f 1` prints, hit for hit, ties included.  2003 sequences: neither a whole number of 16-sequence groups nor of
    8-group blocks.  "0,0" / "0,0,0": context devices on the box's one GPU; "0,1" / "0,1,0": two GPUs, where the box
    has them (skipped otherwise)."""
    from oswald_amd import capi
    if "1" in ids and capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % capi.device_count())
    db = _db_and_queries(tmp_path, 2003, [150, 61, 300, 33], 51)
    ndev = len(ids.split(","))
    common = ["-q", str(tmp_path / "q.fasta"), "-d", db, "-k", "150000"]
    for r in ("15", "1500"):      # on-device lists; and beyond 1024 the score table comes to the host
        one = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", r] + common, capture_output=True, text=True)
        many = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", r, "-f", str(ndev)] + common, capture_output=True, text=True,
                              env=dict(os.environ, OSWALD_DEVICE_IDS=ids))
        assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
        a, b = parse_report(one.stdout), parse_report(many.stdout)
        assert len(a) == 4 and all(len(x["hits"]) == int(r) for x in a)
        assert a == b
        assert f"Number of FPGAs:\t\t{ndev}\n" in many.stdout
    # ... and with every device's first piece cut into a head and the rest (what the tool does from 32 MiB on, so that the
    # rest comes in while the head is searched): the same report, with one device and with several
    cut = dict(os.environ, OSWALD_HIP_SPLIT_BYTES="20000", OSWALD_HIP_DEBUG_SLOW="1")   # (the library cuts, at the upload: oswald_hip_chunk_upload_async)
    one_cut = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", "15"] + common, capture_output=True, text=True, env=cut)
    many_cut = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", "15", "-f", str(ndev)] + common, capture_output=True, text=True,
                              env=dict(cut, OSWALD_DEVICE_IDS=ids))
    ref15 = subprocess.run([hostlib.CLI, "-O", "search", "-m", "0", "-r", "15"] + common, capture_output=True, text=True)
    assert one_cut.returncode == 0 and many_cut.returncode == 0, one_cut.stderr + many_cut.stderr
    assert parse_report(one_cut.stdout) == parse_report(ref15.stdout) == parse_report(many_cut.stdout)
    assert "upload cut in two" in one_cut.stderr and many_cut.stderr.count("upload cut in two") == ndev


def test_cli_chunk_size_is_clamped_to_device_memory(tmp_path):
    """The reference adapts -k to the device's memory in init() (utils.c:162-168).  Here: a -k far beyond what any
    chunk may be is clamped (32-bit offsets inside a chunk) instead of failing in the middle of a search, and on a
    device with little free memory (OSWALD_HIP_FAKE_FREE_MEM, a test hook of oswald_hip_max_chunk_size) the limit the
    footer reports is below the default 128 MiB -- with the same hits either way."""
    db = _db_and_queries(tmp_path, 900, [150, 61, 300], 61)
    common = ["-O", "search", "-m", "0", "-r", "12", "-q", str(tmp_path / "q.fasta"), "-d", db]
    ref = subprocess.run([hostlib.CLI] + common, capture_output=True, text=True)
    big = subprocess.run([hostlib.CLI] + common + ["-k", "100000000000"], capture_output=True, text=True)
    small = subprocess.run([hostlib.CLI] + common, capture_output=True, text=True, env=dict(os.environ, OSWALD_HIP_FAKE_FREE_MEM="3000000000"))
    for p in (ref, big, small):
        assert p.returncode == 0, p.stderr
    limit = lambda p: int(re.search(r"Max. chunk size in FPGA:\t(\d+) bytes", p.stdout).group(1))
    assert limit(ref) == 134217728
    assert 134217728 < limit(big) <= 0xfff00000
    assert 0 < limit(small) < 134217728
    assert parse_report(big.stdout) == parse_report(ref.stdout) == parse_report(small.stdout)
    # the hybrid mode -- the tool's default -- cuts the database by the same clamped limit (the reference clamps in init() for every mode)
    hyb = subprocess.run([hostlib.CLI] + [x if x != "0" else "1" for x in common], capture_output=True, text=True, env=dict(os.environ, OSWALD_HIP_FAKE_FREE_MEM="3000000000"))
    assert hyb.returncode == 0, hyb.stderr
    assert limit(hyb) == limit(small) and parse_report(hyb.stdout) == parse_report(ref.stdout)
    # a device that cannot hold one group of sequences: a message, not a crash
    none = subprocess.run([hostlib.CLI] + common, capture_output=True, text=True, env=dict(os.environ, OSWALD_HIP_FAKE_FREE_MEM="1000000"))
    assert none.returncode != 0 and "max_chunk_size is smaller than one group" in none.stdout + none.stderr
