#!/usr/bin/env python3
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.

Generates the golden vectors under tests/golden/ by running the *compiled
reference* (oracle/_ref/libosw_ref.so, built in place from /root/reference by
`make -C oracle ref`) on inputs defined here.  Only inputs and the reference's
outputs are stored -- no reference source text.  Run in the build container:

    python oracle/gen_golden.py [output directory, default tests/golden]

(tests/test_golden_regenerates.py runs it into a scratch directory wherever the
reference is present and compares the files with the committed ones.)

Vectors:
  G1 alphabet.json ....... preprocess_db on every upper-case letter (+ a few
                           other bytes): byte -> code
  G2 preprocess.json ..... small FASTA files -> exact .info / .seq / .desc bytes
  G2q queries.json ....... load_query_sequences: order, residues, lengths, titles
  G3 layout.npz .......... assemble_multiple_chunks_db: n, nbb, disp, chunking,
                           interleaved bytes (small cases) / digests (large)
  G4 submat.npz .......... the 8 matrices, 768 bytes each
  G5 scores.npz .......... sw_host int32 scores (C1 shape, multi-query PAM250,
                           adversarial saturation cases)
  G6 sort.npz ............ sort_scores permutations (tie order), threads 1/2/4
"""
from __future__ import annotations

import base64
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oswald_amd import dblayout, synth  # noqa: E402  (inputs only: generators and layout of *inputs*)

OUT = os.path.join(ROOT, "tests", "golden")
MATRICES = ("blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250")


def ref_lib():
    path = os.path.join(HERE, "_ref", "libosw_ref.so")
    if not os.path.exists(path):
        raise SystemExit("build the reference first: make -C oracle ref")
    lib = C.CDLL(path)
    lib.ref_submat.restype = C.c_void_p
    lib.ref_load_queries.restype = C.c_ulong
    lib.ref_queries_residues.restype = C.c_void_p
    lib.ref_queries_lengths.restype = C.c_void_p
    lib.ref_queries_disp.restype = C.c_void_p
    lib.ref_queries_title.restype = C.c_char_p
    lib.ref_queries_title.argtypes = [C.c_ulong]
    lib.ref_assemble.restype = C.c_uint
    lib.ref_chunk_groups.restype = C.c_uint
    lib.ref_chunk_accum.restype = C.c_ulong
    lib.ref_chunk_vD.restype = C.c_ulong
    for f in ("ref_chunk_b", "ref_chunk_n", "ref_chunk_nbb", "ref_chunk_disp"):
        getattr(lib, f).restype = C.c_void_p
    lib.ref_header.restype = C.c_char_p
    lib.ref_header.argtypes = [C.c_ulong]
    return lib


def arr(ptr, ctype, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).copy()


def b64(b: bytes) -> str:
    return base64.b64encode(b).decode()


def submat_of(lib, name):
    return arr(lib.ref_submat(name.encode()), C.c_int8, 768)


def sw_host_group(lib, a, b16, n, sm, go, ge, block=256):
    a = np.ascontiguousarray(a, np.uint8)
    b16 = np.ascontiguousarray(b16, np.uint8)
    sm = np.ascontiguousarray(sm, np.int8)
    out = np.zeros(16, np.int32)
    lib.ref_sw_host_group(a.ctypes.data_as(C.c_void_p), C.c_ushort(len(a)), b16.ctypes.data_as(C.c_void_p), C.c_ushort(n),
                          sm.ctypes.data_as(C.c_void_p), go, ge, block, out.ctypes.data_as(C.c_void_p))
    return out


def ref_scores(lib, queries, b, n, disp, sm, go, ge, block=256):
    """[nq][ngroups*16] via the reference's sw_host on every 16-lane group."""
    out = np.zeros((len(queries), len(n) * 16), np.int32)
    for qi, q in enumerate(queries):
        for g in range(len(n)):
            grp = b[int(disp[g]):int(disp[g]) + int(n[g]) * 16]
            out[qi, g * 16:(g + 1) * 16] = sw_host_group(lib, q, grp, int(n[g]), sm, go, ge, block)
    return out


def preprocess(lib, tmp, fasta_text: str, name: str, threads=1):
    src = os.path.join(tmp, name + ".fasta")
    with open(src, "w") as f:
        f.write(fasta_text)
    dst = os.path.join(tmp, name)
    lib.ref_preprocess_db(src.encode(), dst.encode(), threads)
    return {ext: open(dst + "." + ext, "rb").read() for ext in ("info", "seq", "desc")}, dst


FASTA_CASES = {
    # ties in length keep FASTA order; multi-line records; B/Z/X and the removed letters J/O/U
    "ties": ">s1 first\nACDEFGHIK\n>s2 second\nAC\n>s3 third, same length as s1\nLMNPQRSTV\n>s4\nWYBZX\n>s5 tie with s2\nJO\n>s6\nU\n",
    "multiline": ">sp|P1|multi line record\nMKVLAAGIVGLLLAQ\nPSWAHHHHHH\nGG\n>sp|P2|short\nMK\n>sp|P3|one\nW\n>sp|P4|sixty\n" + "ACDEFGHIKLMNPQRSTVWY" * 3 + "\n",
    "alphabet": "".join(f">{c}\n{c}\n" for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZ"),
}


def gen_alphabet(lib, tmp):
    # every upper-case letter as a one-residue sequence; stable sort keeps the order
    out, _ = preprocess(lib, tmp, FASTA_CASES["alphabet"], "alpha")
    codes = np.frombuffer(out["seq"], np.uint8)[26 * 2:]
    table = {chr(ord("A") + i): int(codes[i]) for i in range(26)}
    # a few bytes the reference does not reject (lower case, digits, '*', '-')
    extra = "abcxyz019*-"
    out2, _ = preprocess(lib, tmp, "".join(f">{i}\n{c}\n" for i, c in enumerate(extra)), "alpha2")
    codes2 = np.frombuffer(out2["seq"], np.uint8)[len(extra) * 2:]
    table_extra = {c: int(codes2[i]) for i, c in enumerate(extra)}
    json.dump({"upper": table, "other_bytes": table_extra}, open(os.path.join(OUT, "alphabet.json"), "w"), indent=1)


def strip_uninitialised(lines, fasta_text):
    """The reference keeps every title with ONE uninitialised byte behind it (titles[i][length] = 0 is written one position too
    far, sequences.c:116, :333): its .desc lines and in-memory titles end in a byte that differs from run to run.  It is taken
    off HERE, so that a regenerated fixture is byte-identical to the committed one: a line that is a
    title line of the FASTA text plus one byte loses that byte.  (When the byte happens to be 0 the line has none to lose: how
    many were stripped differs from run to run and is printed, not stored.)  -> (lines, how many were stripped)"""
    titles = {l.encode() for l in fasta_text.split("\n") if l.startswith(">")}
    out, n = [], 0
    for l in lines:
        if l not in titles and l[:-1] in titles:
            l, n = l[:-1], n + 1
        out.append(l)
    return out, n


def gen_preprocess(lib, tmp):
    cases = {}
    for name, text in FASTA_CASES.items():
        for threads in (1, 4):
            out, _ = preprocess(lib, tmp, text, f"{name}_{threads}", threads)
            desc, stripped = strip_uninitialised(out["desc"].split(b"\n"), text)
            cases[f"{name}/threads{threads}"] = {"fasta": text, "threads": threads, "info": out["info"].decode(),
                                                 "seq_b64": b64(out["seq"]), "desc_b64": b64(b"\n".join(desc))}
            print(f"preprocess {name}/threads{threads}: {stripped} uninitialised title bytes stripped")
    json.dump(cases, open(os.path.join(OUT, "preprocess.json"), "w"), indent=1)


def gen_queries(lib, tmp):
    text = (">q_long some description\nMKVLAAGIVGLLLAQPSWAHHHHHHGG\nACDEFGHIK\n>q_short\nMKW\n>q_mid\nACDEFGHIKLMNPQ\n"
            ">q_short2 tie\nJOU\n>q_b\nBZXBZX\n")
    src = os.path.join(tmp, "queries.fasta")
    open(src, "w").write(text)
    Q = C.c_ulong(0)
    nq = lib.ref_load_queries(src.encode(), 2, C.byref(Q))
    m = arr(lib.ref_queries_lengths(), C.c_ushort, nq)
    disp = arr(lib.ref_queries_disp(), C.c_uint, nq + 1)
    a = arr(lib.ref_queries_residues(), C.c_uint8, Q.value)
    titles, stripped = strip_uninitialised([lib.ref_queries_title(i) for i in range(nq)], text)
    titles = [b64(t) for t in titles]
    print(f"queries: {stripped} uninitialised title bytes stripped")
    json.dump({"fasta": text, "nq": int(nq), "Q": int(Q.value), "m": m.tolist(), "disp": disp.tolist(), "a": a.tolist(), "titles_b64": titles},
              open(os.path.join(OUT, "queries.json"), "w"), indent=1)


def write_db(lib, tmp, name, seqs):
    fasta = os.path.join(tmp, name + ".fasta")
    synth.write_fasta(fasta, seqs)
    dst = os.path.join(tmp, name)
    lib.ref_preprocess_db(fasta.encode(), dst.encode(), 2)
    return dst


def assemble(lib, dbname, W, max_chunk, ndev):
    out8 = (C.c_ulong * 8)()
    cc = lib.ref_assemble(dbname.encode(), W, C.c_ulong(max_chunk), ndev, 2, out8)
    res = {"seqs": int(out8[0]), "D": int(out8[1]), "maxlen": int(out8[2]), "maxtitle": int(out8[3]), "vgroups": int(out8[4]),
           "vD": int(out8[5]), "max_chunk_vD": int(out8[6]), "chunk_count": int(cc), "chunks": []}
    for c in range(cc):
        g = lib.ref_chunk_groups(c)
        vD = lib.ref_chunk_vD(c)
        res["chunks"].append({
            "groups": int(g), "accum": int(lib.ref_chunk_accum(c)), "vD": int(vD),
            "n": arr(lib.ref_chunk_n(c), C.c_ushort, g), "nbb": arr(lib.ref_chunk_nbb(c), C.c_ushort, g),
            "disp": arr(lib.ref_chunk_disp(c), C.c_uint, g), "b": arr(lib.ref_chunk_b(c), C.c_uint8, vD)})
    return res


def gen_layout(lib, tmp):
    store = {}
    meta = {}
    for nseq in (1, 15, 16, 17, 1000):
        L, R, O = synth.make_database(nseq, seed=4242 + nseq)
        seqs = [R[O[i]:O[i + 1]] for i in range(nseq)]
        db = write_db(lib, tmp, f"lay{nseq}", seqs)
        configs = [("k128M_f1", 134217728, 1)]
        if nseq == 1000:
            configs += [("k200000_f1", 200000, 1), ("k128M_f4", 134217728, 4)]
        for tag, k, f in configs:
            r = assemble(lib, db, 16, k, f)
            key = f"n{nseq}/{tag}"
            meta[key] = {k2: r[k2] for k2 in ("seqs", "D", "maxlen", "maxtitle", "vgroups", "vD", "max_chunk_vD", "chunk_count")}
            meta[key].update({"nseq": nseq, "seed": 4242 + nseq, "max_chunk": k, "ndev": f,
                              "chunk_groups": [c["groups"] for c in r["chunks"]], "chunk_accum": [c["accum"] for c in r["chunks"]],
                              "chunk_vD": [c["vD"] for c in r["chunks"]],
                              "b_sha256": [hashlib.sha256(c["b"].tobytes()).hexdigest() for c in r["chunks"]]})
            for ci, c in enumerate(r["chunks"]):
                store[f"{key}/c{ci}/n"] = c["n"]
                store[f"{key}/c{ci}/nbb"] = c["nbb"]
                store[f"{key}/c{ci}/disp"] = c["disp"]
                if nseq <= 17:
                    store[f"{key}/c{ci}/b"] = c["b"]
        # the headers as the reference reads them back (for the report)
        if nseq == 17:
            lib.ref_load_headers(db.encode(), C.c_ulong(nseq), meta[f"n{nseq}/k128M_f1"]["maxtitle"])
            heads, stripped = strip_uninitialised([lib.ref_header(i).rstrip(b"\n") for i in range(nseq)], open(db + ".fasta").read())
            print(f"headers: {stripped} uninitialised title bytes stripped")
            meta["n17/headers_b64"] = [b64(h + b"\n") for h in heads]
    np.savez_compressed(os.path.join(OUT, "layout.npz"), **store)
    json.dump(meta, open(os.path.join(OUT, "layout.json"), "w"), indent=1)


def gen_submat(lib):
    np.savez_compressed(os.path.join(OUT, "submat.npz"), **{n: submat_of(lib, n) for n in MATRICES})


def gen_scores(lib):
    store = {}
    # (i) C1 shape: one query m = 375 against 1000 synthetic sequences, BLOSUM62 10/2
    q1 = synth.make_queries([375])
    L, R, O = synth.make_database(1000, q1, homologs_per_query=12)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    sm = submat_of(lib, "blosum62")
    store["c1/scores"] = ref_scores(lib, q1, b, n, disp, sm, 10, 2)
    store["c1/b_sha256"] = np.frombuffer(hashlib.sha256(b.tobytes()).digest(), np.uint8)
    # (ii) 20 queries x 2000 sequences, PAM250 14/2, block width 100 (crosses blocks)
    q2 = synth.make_queries(synth.default_query_lengths())
    L, R, O = synth.make_database(2000, q2, seed=77, homologs_per_query=3)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    sm = submat_of(lib, "pam250")
    store["multi/scores"] = ref_scores(lib, q2, b, n, disp, sm, 14, 2, block=100)
    store["multi/b_sha256"] = np.frombuffer(hashlib.sha256(b.tobytes()).digest(), np.uint8)
    # (iii) adversarial: all-W runs around the int8 and int16 ceilings (W-W = 11 with BLOSUM62),
    # length-1 sequences, a 5 %-mutated copy of an m = 3100 query (score > 32767)
    w = synth.ALPHABET.index("W")
    qa = np.full(3100, w, np.uint8)
    qb = synth.make_queries([3100], seed=909)[0]
    lens = [1, 2, 11, 12, 2977, 2978, 2979, 2980, 3100]
    seqs = [np.full(k, w, np.uint8) for k in lens] + [synth.mutate(qb, 0.05, 5), np.array([0], np.uint8), qb[:50].copy()]
    lengths = np.array([len(s) for s in seqs], np.uint16)
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum(lengths, out=off[1:])
    order, sl, sr, so = dblayout.sort_by_length(lengths, np.concatenate(seqs), off)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    sm = submat_of(lib, "blosum62")
    store["adv/scores"] = ref_scores(lib, [qa, qb, qa[:12], qa[:1]], b, n, disp, sm, 10, 2)
    store["adv/b"] = b
    store["adv/n"] = n
    store["adv/disp"] = disp.astype(np.uint32)
    store["adv/sorted_lengths"] = sl
    np.savez_compressed(os.path.join(OUT, "scores.npz"), **store)


def gen_sort(lib):
    store = {}
    rng = np.random.default_rng(2016)
    for size in (1, 2, 3, 11, 1000):
        sc = rng.integers(0, 8 if size > 3 else 2, size).astype(np.int32)
        store[f"s{size}/in"] = sc
        for threads in (1, 2, 4):
            s2 = sc.copy()
            ix = np.zeros(size, np.uint32)
            lib.ref_sort_scores(s2.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), C.c_ulong(size), threads)
            store[f"s{size}/t{threads}/sorted"] = s2
            store[f"s{size}/t{threads}/index"] = ix
    np.savez_compressed(os.path.join(OUT, "sort.npz"), **store)


def main():
    global OUT
    if len(sys.argv) > 1:
        OUT = os.path.abspath(sys.argv[1])
    os.makedirs(OUT, exist_ok=True)
    lib = ref_lib()
    with tempfile.TemporaryDirectory() as tmp:
        gen_alphabet(lib, tmp)
        gen_preprocess(lib, tmp)
        gen_queries(lib, tmp)
        gen_layout(lib, tmp)
    gen_submat(lib)
    gen_scores(lib)
    gen_sort(lib)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
