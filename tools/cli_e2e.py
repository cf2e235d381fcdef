#!/usr/bin/env python3
"""tools/cli_e2e.py [nseq [tmpdir [query lengths, comma separated]]] -- end-to-end wall time of the `oswald` CLI on a synthetic database
(default: the 20 queries of C2; "375" = Q1, "5000" = C5); OSWALD_E2E_ARGS: further arguments of the search, e.g. "-k 33554432"."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oswald_amd import synth
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp/osw_e2e"
os.makedirs(tmp, exist_ok=True)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oswald_amd", "oswald")
qs = synth.make_queries([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else synth.default_query_lengths())
L, R, O = synth.make_database(nseq, qs)
t = time.time(); synth.write_fasta(f"{tmp}/db.fasta", [R[O[i]:O[i+1]] for i in range(nseq)]); synth.write_fasta(f"{tmp}/q.fasta", qs); print("fasta written", round(time.time()-t,1), "s")
t = time.time(); subprocess.run([cli, "-O", "preprocess", "-i", f"{tmp}/db.fasta", "-o", f"{tmp}/db"], check=True, stdout=subprocess.DEVNULL); print("preprocess", round(time.time()-t,2), "s")
env = dict(os.environ, OSWALD_DEBUG_PHASES="1")
for label, extra in (("group cache", {}), ("interleave from .seq", {"OSWALD_NO_GROUP_CACHE": "1"})):
    t = time.time(); p = subprocess.run([cli, "-O", "search", "-m", "0", "-q", f"{tmp}/q.fasta", "-d", f"{tmp}/db"] + os.environ.get("OSWALD_E2E_ARGS", "").split(), capture_output=True, text=True, env=dict(env, **extra)); print(f"search wall ({label})", round(time.time()-t,2), "s rc", p.returncode)
    print(p.stderr)
    print("\n".join(l for l in p.stdout.split("\n") if l.startswith(("Search time", "Search speed", "Database size"))))
print(p.stdout.split("Query no.")[1][:400] if "Query no." in p.stdout else p.stderr[-500:])
