#!/usr/bin/env python3
"""bench.py -- GCUPS of the MI355X Smith-Waterman search path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every query of the
workload against the rank's resident part of the database (DP kernels incl. the
exact int32 re-run of saturated cells, on-device top-r per chunk) and, for
N > 1, the gather of the per-GPU top-r lists over RCCL.  Inputs (re-tiled
residues, query profiles) are RESIDENT IN HBM before the timed region starts
(the bench contract of this build fixes `value` to that region).  The reference's
own timed region (FPGAsearch.c:80-276: upload + kernels + download of
the score tables, the host buffers handed over inside the clock) is measured in
the same run with the same rigour -- its own warm-up passes, K timed steps between
barriers -- and reported next to it as `inclusive` (`value_inclusive` at the top
level); on the default workload the two are within half a per cent.

Workload (default, every N): BASELINE.json configs[3] (C4), the configuration the
  metric "GCUPS at 1/2/4/8 MI355X" is quoted on -- 20 queries of length 100..1000
  (sum 11 000) against ONE 1 000 000-sequence synthetic length-binned database
  (~364.6 M residues), BLOSUM62, gap 10/2, int16 cells (packed, exact below
  22256; sequences above are re-run in int32).  It fits one GPU, so N = 1 runs the
  same database as N = 2, 4, 8: one series, total work fixed as N grows
  ("scaling": "strong").  The database is length-sorted once and divided among
  the ranks (--shard-rule):
    deal       (default) wave blocks of 128 consecutive sorted sequences dealt to
               the ranks in alternating order (oswald_amd/multigpu.py): every
               rank gets the same length distribution and they finish together;
    reference  the reference's rule: contiguous shards of ceil(vD/ndev) padded
               residues (host/src/sequences.c:510-515), chunk c to device c mod
               ndev (host/src/FPGAsearch.c:132-138) -- the longest sequences all
               land in the last shard.
  Database indices are positions in the globally sorted database, so the merged
  top-10 is the same list for every N and rule (tests/reference_runs/bench_top_*.json
  holds a single-GPU run's list -- written by the HIP path, its scores pinned to
  the oracle by tests/test_reference_runs.py; a mismatch fails the run).  At N = 1
  the list of THIS run is checked against the oracle itself: `top10_equals_oracle`.
  --nseq overrides the TOTAL number of database sequences (--nseq 100000 =
  BASELINE.json configs[1], C2); --workload c3 / c5 / q1 default to 100 000;
  --weak restores round 1's mode (an independent --nseq database per GPU).

GCUPS = sum(query lengths) x unpadded database residues / seconds / 1e9, the
reference's definition (reference host/src/FPGAsearch.c:324).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9                 # max clock (MI355X_MICROARCH.md)
N_CU = 256
SIMD_PER_CU = 4
# VALU instructions per wave per query row (= per 128 cells) of the DP kernels: {first-pass arithmetic: (one query per
# lane: two sequences per lane, their scores paired and added by one v_pk_mad_i16; query pairs)}
PK_OPS_PER_ROW = {16: (6.5, 6.5), 32: (13.0, 13.0), 8: (20.0, 20.0)}  # 8: 39 SWAR instructions + 1 v_perm_b32 per row of a 2 x 2 tile = 256 cells (q8_cell.h, CellQ8F)
# ... and what such a row costs in core-clock cycles per SIMD at 4 waves per SIMD.  int16: MEASURED on the cell's own
# instruction mix (tools/oprate2.hip, profiles/r02_oprate2_valu_mix.txt: 3.5 VOP3P + 3 VOP2 = 25.3 cycles for the query-pair
# row; the sequence-pair row -- since the third session of round 4 one v_pk_mad_i16 pairs the two sequences' scores and adds them, where a
# v_perm_b32 and a packed add did (7.5 instructions, 30.5 cycles) -- is the same row with a VOP3P instead of the 32-bit add: 27.3 cycles
# without its loads, tools/oprate8.hip "none", profiles/r04_oprate8_pk_mad.txt); int8: MEASURED on the cell's own column
# step at the six waves per SIMD osw_sw_q8 runs at (tools/oprate_q8.hip, profiles/r04_oprate_q8.txt: the hand-scheduled cell of
# round 4 takes 90.9 cycles per row of a 2 x 2 tile = 256 cells; round 3's compiler-scheduled one took 129.5); int32 (hand-scheduled since
# round 5: 6.5 instructions per row of 64 cells): MEASURED, tools/oprate9.hip "none" at three waves per SIMD, 26.37 cycles per 64 cells (profiles/r05_oprate9_int32_row.txt).
ROW_CYCLES = {16: (27.3, 25.3), 32: (52.74, 52.74), 8: (45.45, 45.45)}
DTYPE = {16: "int16", 32: "int32", 8: "int8"}
CELL_LABEL = {16: "int16 cells (packed, column frames, exact < 22256), int32 re-run above", 32: "int32 cells",
              8: "int8 cells (four 7-bit SWAR cells per register) with int16 re-run of what leaves their range, int32 above"}
KERNEL_SOURCES = ("oswald_amd/csrc/sw_kernels.hip", "oswald_amd/csrc/sw_kernels.h", "oswald_amd/csrc/q8_cell.h", "oswald_amd/csrc/osw_planner.inc")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=0, help="TOTAL database sequences (default: 1000000 = C4 for the default workload at every --gpus, 100000 for c3 / c5 / q1; 100000 = C2); per GPU with --weak")
    ap.add_argument("--shard-rule", default="deal", choices=["deal", "reference"], help="how the sorted database is divided among the ranks (see the module docstring)")
    ap.add_argument("--weak", action="store_true", help="round-1 mode: every rank searches its own independent --nseq database")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "q1", "hi", "hi8"],
                    help="c2 / c3 / c5 / q1: BASELINE configs; hi: escalation-heavy (ten queries of 4400-6600 residues against a database in which 1 %% of the "
                         "sequences are near-copies of them: scores beyond the int16 cells, re-run in int32); hi8: the C3 set on the 8-bit cells with 5 %% of the "
                         "database planted homologs (pairs that leave the 7-bit range, re-run in int16)")
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--max-chunk", type=int, default=134217728, help="chunk size limit in bytes (the reference's -k, default 128 MiB)")
    ap.add_argument("--cell-bits", type=int, default=0, choices=[0, 8, 16, 32],
                    help="cell arithmetic: 16 = packed int16 (the cells BASELINE.json configs[1] names; the library's default), 32 = int32 only, "
                         "8 = SWAR 8-bit first pass with int16 re-run (configs[2]); default: 8 for --workload c3 (the cell mode that configuration names), else 16")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 = skip)")
    ap.add_argument("--cpu-lanes", type=int, default=32, choices=[16, 32], help="16 = SSE4.1 port, 32 = AVX2 port")
    ap.add_argument("--gather", default="auto", choices=["auto", "lib", "torch"],
                    help="who carries the top-r gather between the ranks: lib = the C ABI itself (ncclAllGather inside oswald_hip_topr on a communicator "
                         "made by oswald_hip_comm_init_rank; torch.distributed only hands the id round and keeps time), torch = torch.distributed.all_gather "
                         "of the ranks' lists; auto = lib over RCCL, torch for the gloo rehearsal")
    ap.add_argument("--per-chunk-launches", action="store_true", help="the resident steps search the rank's chunks one by one (oswald_hip_chunk_search, rounds 1-5) instead of as one "
                                                                    "launch over all of them (oswald_hip_search_resident)")
    ap.add_argument("--comm", action="store_true", help="N = 1: give the one rank a process-level RCCL communicator all the same (the gather path of N > 1 at world size 1)")
    ap.add_argument("--write-top-reference-run", action="store_true", help="N = 1 only: write tests/reference_runs/bench_top_<workload>_<nseq>.json (this run's merged top list)")
    return ap.parse_args()


def workload(name):
    if name == "c2":
        return dict(qlens=None, matrix="blosum62", go=10, ge=2, label="20 queries len 100-1000 x {nseq}-seq synthetic DB, BLOSUM62 10/2")
    if name == "c3":
        return dict(qlens=None, matrix="pam250", go=14, ge=2, label="20 queries len 100-1000 x {nseq}-seq synthetic DB, PAM250 14/2")
    if name == "hi":
        # every query's self-score is beyond the column-frame cell's ceiling (22 256), the longer ones' beyond the plain int16
        # cell's (30 576): ~5.3 per residue.  Copies at 1 .. 30 % substitutions: most of them above 22 256, the best above 30 576.
        return dict(qlens=[4400 + 245 * k for k in range(10)], matrix="blosum62", go=10, ge=2, planted_share=0.01,
                    rates=[0.01, 0.02, 0.03, 0.05, 0.07, 0.10, 0.14, 0.18, 0.24, 0.30],
                    label="10 queries len 4400-6605 x {nseq}-seq synthetic DB of which 1 % are near-copies of the queries (1-30 % substitutions), BLOSUM62 10/2")
    if name == "hi8":
        return dict(qlens=None, matrix="pam250", go=14, ge=2, planted_share=0.05, rates=None,
                    label="20 queries len 100-1000 x {nseq}-seq synthetic DB of which 5 % are planted homologs (5-60 % substitutions), PAM250 14/2")
    if name == "q1":
        return dict(qlens=[375], matrix="blosum62", go=10, ge=2, label="1 query len 375 (the C1 query) x {nseq}-seq synthetic DB, BLOSUM62 10/2")
    return dict(qlens=[5000], matrix="blosum62", go=10, ge=2, label="1 query len 5000 x {nseq}-seq synthetic DB, BLOSUM62 10/2")


def source_digest():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # this pool's host driver supports dmabuf IPC only: without it RCCL's exchange of buffers between the ranks' processes fails
    # (hipIpcGetMemHandle: invalid argument).  The image exports it; a launcher that built its own environment may not have.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        # the library's bring-up ends with a warm-up kernel that keeps every CU busy until the first real work: meant for a
        # one-shot tool's first search; here the warm-up steps do that job, and the ranks' collectives' own bring-up should find the GPUs free
        os.environ.setdefault("OSWALD_HIP_WARM_MS", "0")
    if world == 1 and args.gpus > 1:
        # started without a launcher: start the ranks ourselves (before anything touches the GPU) and leave with their code
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # N = 1: torch never touches the GPU.  With torch's runtime (its streams, its caching allocators) in the process the
    # upload-inclusive pass of a SHORT search took 7 - 9 ms in most processes instead of 2.9 (LABNOTES round 4 (6): fast through the C
    # ABI alone, fast under rocprofv3, slow with torch whatever the library's stream classes) -- the library's streams then share the
    # runtime's few hardware queues with torch's.  torch.distributed is what N > 1 needs torch for (rendezvous, barrier, MAX of the
    # elapsed times); a single rank needs none of it: its barrier is the library's own wait for every stream it has.
    from oswald_amd import capi, multigpu
    torch = None
    if world > 1:
        # (torch first: it brings its own copy of the HIP runtime, and a process in which the library has loaded the system's copy
        # before torch initialises finds "no HIP GPUs" in torch; loaded second, the library shares torch's)
        import torch
        ngpu_visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    else:
        ngpu_visible = capi.device_count()
    if ngpu_visible <= 0:
        raise SystemExit("bench.py needs a GPU: the search path is HIP only (no CPU fallback)")
    # one rank per GPU over RCCL ("nccl").  OSWALD_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks, to be asked for by name: ranks share the visible GPUs and the top-r gather runs over gloo on host tensors.
    # A failed RCCL bring-up ends the run non-zero (multigpu.init_collective): there is no fallback.
    backend = os.environ.get("OSWALD_BENCH_BACKEND", "nccl")
    gpu = local_rank % ngpu_visible
    dist = None
    dev = coll_dev = None
    if world > 1:
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
        if backend == "nccl" and world > torch.cuda.device_count():
            raise SystemExit(f"bench.py: {world} ranks over RCCL need a GPU each, {torch.cuda.device_count()} visible "
                             "(OSWALD_BENCH_BACKEND=gloo rehearses the sharding with ranks sharing a GPU)")
        dist = multigpu.init_collective(backend, dev)
        coll_dev = dev if backend == "nccl" else torch.device("cpu")
    gather = args.gather if args.gather != "auto" else ("lib" if (world > 1 and backend == "nccl") or args.comm else "torch")
    if gather == "lib" and world > 1 and backend != "nccl":
        raise SystemExit("--gather lib needs one GPU per rank (RCCL); the gloo rehearsal gathers through torch.distributed")

    from oswald_amd import dblayout, submat, synth

    wl = workload(args.workload)
    qlens = wl["qlens"] or synth.default_query_lengths()
    queries = synth.make_queries(qlens)
    sm = submat.load(wl["matrix"])
    m = np.array(qlens, dtype=np.uint16)
    a_disp = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
    a = np.concatenate(queries)
    sum_m = int(m.astype(np.int64).sum())
    nq = len(qlens)

    # The database: a plan of the WHOLE database on every rank (lengths and planted homologs only), sorted by
    # length once, cut into chunks by the reference's rule; every rank materialises and interleaves only the
    # sequences of its own chunks, exactly like the reference's preprocessed + assembled database (W = 16,
    # group lengths padded to x28).
    t0 = time.time()
    strong = not args.weak
    nseq_total = args.nseq or (1000000 if args.workload == "c2" and not args.weak else 100000)
    per_query = max(1, int(round(wl["planted_share"] * nseq_total / nq))) if wl.get("planted_share") else 12   # planted copies per query
    if strong:
        plan = synth.DatabasePlan(nseq_total, queries, synth.SEED_DB, per_query, wl.get("rates"))
        shard_world, shard_rank = world, rank
    else:
        plan = synth.DatabasePlan(nseq_total, queries, synth.SEED_DB + 1000003 * rank, per_query, wl.get("rates"))
        shard_world, shard_rank = 1, 0
    shard = multigpu.ShardedDatabase(plan, 16, args.max_chunk, shard_world, shard_rank, args.shard_rule)
    index_base = 0 if strong else rank * nseq_total      # weak mode: global index = shard base + sorted position

    if world > 1 and backend != "nccl":
        # rehearsal: several PROCESSES on one GPU.  The library's DMA streams sit in priority classes of their own (so that a copy
        # never queues behind a persistent search grid in a shared hardware queue); with four processes' queues of three classes
        # on one GPU the scheduler alternates badly between the processes (550 instead of 340 ms per step at four ranks,
        # tools/ab_gloo4.sh): a configuration that exists for this rehearsal only, so the classes are switched off in it.
        os.environ.setdefault("OSWALD_HIP_NO_STREAM_CLASSES", "1")
    ctx = capi.Context(1, [gpu])
    gather_note = None
    if gather == "lib":
        # the gather lives in the C ABI: rank 0 makes the id, torch.distributed hands it round, every rank joins
        lib_error = None
        try:
            ident = [capi.comm_unique_id() if rank == 0 else None]
        except capi.OswaldHipError as e:
            ident, lib_error = [None], str(e)
        if dist is not None:
            dist.broadcast_object_list(ident, src=0, device=coll_dev)
        if ident[0] is None:
            lib_error = lib_error or "rank 0 could not make a communicator id"
        elif os.environ.get("OSWALD_BENCH_FAIL_LIB_COMM") in ("1", "all"):   # test hook: no rank joins (tests/test_gpu_bench_contract.py)
            lib_error = "OSWALD_BENCH_FAIL_LIB_COMM is set"
        else:
            try:
                ctx.comm_init_rank(ident[0], world, rank)
            except capi.OswaldHipError as e:
                lib_error = str(e)
            # test hook "rank:<n>": rank n reports a failure AFTER the (collective) bring-up -- the partial case: every other
            # rank holds a working communicator at this point, and the joint decision below must take it away from them
            if os.environ.get("OSWALD_BENCH_FAIL_LIB_COMM") == f"rank:{rank}":
                lib_error = lib_error or "OSWALD_BENCH_FAIL_LIB_COMM names this rank"
        # Whether the library's communicator stands is decided by ALL ranks together (a flag through the RCCL group torch
        # already has): either every rank gathers inside the C ABI or every rank gathers through torch.distributed --
        # over RCCL in both cases, never over another backend.  `--gather lib` asked for by name does not degrade.
        failed = 1 if lib_error else 0
        if dist is not None:
            flag = torch.tensor([failed], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = int(flag.item())
        if failed:
            why = lib_error or "another rank could not join the library's communicator"
            if args.gather == "lib" or dist is None:
                raise SystemExit(f"bench.py: the library's RCCL communicator could not be made on rank {rank}: {why}")
            # every rank gives up what it may hold of the library's communicator: a rank that did join would otherwise still
            # all-gather over it inside oswald_hip_topr -- with ranks that never joined
            ctx.comm_destroy()
            gather = "torch"
            gather_note = ("oswald_hip_comm_init_rank failed (" + why[:200] + "): the ranks' lists are carried by torch.distributed.all_gather "
                           "over the same RCCL backend instead of by oswald_hip_topr")
            if rank == 0:
                print("bench.py: " + gather_note, file=sys.stderr, flush=True)
    cell_bits = args.cell_bits or (8 if args.workload in ("c3", "hi8") else 16)
    ctx.set_scoring(sm, wl["go"], wl["ge"], cell_bits)
    ctx.set_queries(a, m, a_disp)
    chunks = []          # resident chunks of this rank (+ their host arrays for the PCIe-inclusive leg)
    d_local = 0
    for k in range(len(shard.mine)):
        c = shard.chunk(k)
        if c["b"].size >= 2**32:
            raise SystemExit("chunk too large; lower --max-chunk")
        c["h"] = ctx.chunk_upload(c["b"], c["n"], c["disp"], 16)
        # the chunk's sequences -> positions in the sorted database (--weak: every rank's database has its own range)
        ctx.chunk_set_index(c["h"], 0, c["nseq"], c["gpos"] + (index_base if gather == "lib" else 0))
        chunks.append(c)
        d_local += int(c["off"][-1])
    t_gen = time.time() - t0
    geoms = [ctx.chunk_geometry(c["h"]) for c in chunks]

    def barrier():
        ctx.wait()                        # every stream the library has on this rank's GPU (N = 1: the whole barrier)
        if dist is not None:
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)

    # The database is RESIDENT in these steps: the rank's chunks are searched as ONE launch (oswald_hip_search_resident, round 6: every
    # launch boundary costs a ramp, a ragged end and, for short launches, clock); --per-chunk-launches: one by one, as a caller that
    # streams chunks in must (the `inclusive` leg and the command-line tool do)
    one_launch = len(chunks) > 1 and not args.per_chunk_launches

    def search_all():
        if one_launch:
            ctx.search_resident([c["h"] for c in chunks])
        else:
            for c in chunks:
                ctx.chunk_search(c["h"], None)

    def step():
        # every search queues the selection of its chunks' top r behind it and folds it into the GPU's running list
        # (oswald_hip_topr_begin); oswald_hip_topr waits for the device once.  gather == "lib": it also all-gathers the
        # ranks' lists over RCCL and folds them on the GPU -- the list it returns is the job's; "torch": it returns the
        # rank's list and torch.distributed.all_gather carries the lists between the ranks
        ctx.topr_begin(args.top)
        if gather == "lib":
            search_all()
            sc, ix = ctx.topr(args.top)
            return sc, np.where(ix == 0xFFFFFFFF, -1, ix.astype(np.int64))
        search_all()
        return multigpu.rank_step(chunks, lambda c: None, None, nq, args.top, index_base, dist,
                                  coll_dev if dist is not None else None, collect_rank=lambda: ctx.topr(args.top))

    coll_ranks = None
    if dist is not None:  # the first collective of a process group sets up its channels: not part of any step
        seen = torch.zeros(1, dtype=torch.int64, device=coll_dev) + 1
        dist.all_reduce(seen)                      # how many ranks the torch.distributed group really carries
        coll_ranks = int(seen.item())
        if gather == "torch":
            multigpu.gather_topr(np.full((nq, args.top), -1, np.int32), np.full((nq, args.top), -1, np.int64), args.top, dist, coll_dev)
    if gather == "lib":
        coll_ranks = ctx.comm_info()["process_ranks"]   # ... and how many the library's communicator reports (ncclCommCount)
    if world > 1 and coll_ranks != world:
        raise SystemExit(f"bench.py: the collective carries {coll_ranks} ranks, the job has {world}")
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    ctx.kernel_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    top = None
    for _ in range(args.steps):
        top = step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches, rerun = ctx.kernel_stats()
    rerun16_ms, rerun32_ms = ctx.rerun_stats()
    ctx.set_profiling(False)

    elapsed_rank = elapsed
    d_total = float(d_local)
    rank_info = None
    if dist is not None:
        t_all = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        d_all = torch.tensor([d_local], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(d_all, op=dist.ReduceOp.SUM)
        elapsed = float(t_all.item())
        d_total = float(d_all.item())
        # what every rank ran on and how long it took: the first line from a real multi-GPU node explains itself
        mine = {"rank": rank, "local_rank": local_rank, "device": gpu, "device_count": ngpu_visible, "pci_bus_id": pci_bus_id(ctx),
                "collective_ranks": coll_ranks, "shard_sequences": int(sum(c["nseq"] for c in chunks)), "shard_residues": int(d_local),
                "chunks": len(chunks), "ms_per_step": round(elapsed_rank / args.steps * 1e3, 3)}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)

    result = None
    if rank == 0:
        cells_per_step = sum_m * d_total
        gcups = cells_per_step * args.steps / elapsed / 1e9
        # roofline of the dominant kernels (one event pair per chunk search = per launch group)
        nlaunch = max(1, kern_launches)
        alg_step = sum(nq * (g["residue_bytes_per_query"] + 4 * c["nseq"]) + sum_m + 768 for g, c in zip(geoms, chunks))
        alg_bytes = alg_step * args.steps / nlaunch
        kern_s = kern_ms / nlaunch / 1e3
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        kern_gcups = sum_m * d_local * args.steps / (kern_ms / 1e3) / 1e9 if kern_ms > 0 else 0.0
        ops_row = PK_OPS_PER_ROW[cell_bits][1 if nq > 1 else 0]  # a multi-query search runs (mostly) as query pairs
        row_cycles = ROW_CYCLES[cell_bits][1 if nq > 1 else 0]
        valu_ceiling = N_CU * SIMD_PER_CU * (CLOCK_HZ / row_cycles) * 128.0 / 1e9
        kname = {16: "osw_sw_s16qt|osw_sw_s16q+osw_sw_s16(+osw_sw_i32r)", 32: "osw_sw_i32", 8: "osw_sw_q8+osw_sw_pk16(+osw_sw_i32)"}[cell_bits]
        traffic, traffic_note = measured_traffic(args.workload, nseq_total if world == 1 else None, DTYPE[cell_bits], nlaunch / max(1, args.steps))
        cfg_name = {"c2": "C2" if nseq_total == 100000 and world == 1 else "C4" if nseq_total == 1000000 else "C2-shaped", "c3": "C3", "c5": "C5", "q1": "Q1",
                    "hi": "escalation-heavy (int16 -> int32)", "hi8": "escalation-heavy (int8 -> int16)"}[args.workload]
        if cfg_name == "C4" and world == 1:
            cfg_name = "C4 database on one GPU"
        rule_note = "128-sequence wave blocks dealt to the GPUs in alternating order" if args.shard_rule == "deal" else f"the reference's chunk rule (chunk c -> GPU c mod {world})"
        shard_note = (f"one database sharded over {world} GPUs by {rule_note}, "
                      f"{'RCCL' if backend == 'nccl' else backend} all_gather of top-{args.top} ({'inside the C ABI: oswald_hip_topr' if gather == 'lib' else 'torch.distributed'})") if world > 1 and strong else \
                     (f"independent {nseq_total}-sequence database per GPU x{world}, {'RCCL' if backend == 'nccl' else backend} all_gather of top-{args.top}" if world > 1 else "single GPU")
        result = {
            "metric": "GCUPS", "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "timed_region": "search (DP kernels + int32 re-run) + device top-r + merge" + (" + all_gather" if world > 1 else "") + "; database resident in HBM (the bench contract's region); the reference's own timed region, SURVEY 8d (upload + kernels + download, FPGAsearch.c:80-276), is `inclusive` / `value_inclusive`, measured in the same run over its own timed steps",
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": DTYPE[cell_bits], "data": "synthetic",
            "config": {"workload": f"{cfg_name}: " + wl["label"].format(nseq=nseq_total) + ", " + CELL_LABEL[cell_bits] + "; database resident in HBM (re-tiled) before the timed region",
                       "queries": nq, "query_residues": sum_m, "db_sequences_total": nseq_total * (1 if strong else world),
                       "db_residues_total": int(d_total), "matrix": wl["matrix"], "gap_open": wl["go"], "gap_extend": wl["ge"],
                       "top": args.top, "sharding": shard_note, "shard_rule": args.shard_rule, "collective_backend": ("RCCL (nccl)" if backend == "nccl" else backend) if world > 1 or gather == "lib" else None,
                       "collective_note": gather_note, "collective_ranks": coll_ranks,
                       "collective_via": ("liboswald_hip.so: ncclAllGather of nq x r tagged keys inside oswald_hip_topr, folded on the GPU (RCCL %d)" % ctx.comm_info()["rccl_version"]) if gather == "lib"
                                         else ("torch.distributed.all_gather" if world > 1 else None), "chunks_rank0": len(chunks), "max_chunk_bytes": args.max_chunk,
                       "resident_search": ("one launch over the rank's %d resident chunks (oswald_hip_search_resident)" % len(chunks)) if one_launch else "one launch per chunk (oswald_hip_chunk_search)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_ratio": round(traffic / alg_bytes, 2) if traffic and alg_bytes else None,   # counter bytes over algorithmic bytes: what the strips spill and read back
                         "kernel": kname, "kernel_ms": round(kern_s * 1e3, 3), "kernel_gcups": round(kern_gcups, 1),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "launches_per_step": round(nlaunch / max(1, args.steps), 2),
                         "valu": {"ceiling_gcups": round(valu_ceiling, 0), "frac": round(kern_gcups / valu_ceiling, 4),
                                  "instr_per_128_cells": ops_row, "cycles_per_128_cells": row_cycles,
                                  "note": "the DP is VALU-issue bound: instr_per_128_cells VALU instructions per wave per query row (query-pair cell for a multi-query search) issue in cycles_per_128_cells core-clock cycles per SIMD (measured on the cell's instruction mix: profiles/r02_oprate2_valu_mix.txt for the int16 cells, profiles/r04_oprate_q8.txt for the 8-bit cell, profiles/r05_oprate9_int32_row.txt for the int32 cell); ceiling = 1024 SIMDs x 2.4 GHz / that x 128"}},
            # the north star's ">= 0.5 x HBM roofline" is only well posed under the reference's own traffic
            # model, 1 B of substitution score per cell streamed from device DRAM (sw.cl:57): 8 TB/s = 8000 GCUPS
            "reference_traffic_model": {"bytes_per_cell": 1.0, "roofline_gcups": HBM_PEAK_GBS, "frac": round(gcups / world / HBM_PEAK_GBS, 4)},
            "rerun_items_int32": int(rerun), "rerun_items_int16": int(ctx.rerun_counts()[0]),
            # device time of the escalation tiers per step on rank 0 (HIP events around their launches; part of kernel_ms): the int16
            # re-run of what left the 8-bit cells, the exact int32 re-run of what reached the int16 cells' ceiling
            "rerun_ms_per_step": {"int16": round(rerun16_ms / args.steps, 3), "int32": round(rerun32_ms / args.steps, 3),
                                  "share_of_kernel_time": round((rerun16_ms + rerun32_ms) / kern_ms, 4) if kern_ms > 0 else None}, "work_items": int(sum(ctx.chunk_geometry(c["h"])["work_items"] for c in chunks)),
            "max_log2_geometry": int(max([ctx.chunk_geometry(c["h"])["max_log2_geometry"] for c in chunks] or [0])),
            "planned_spill_bytes_per_step": int(sum(ctx.chunk_geometry(c["h"])["planned_spill_bytes"] for c in chunks)),
            "top1_scores": [int(x) for x in top[0][:, 0]] if top is not None else None,
            "setup_s": round(t_gen, 1),
        }
        if rank_info is not None:
            steps_ms = [r["ms_per_step"] for r in rank_info]
            result["ranks"] = rank_info
            result["rank_ms_per_step"] = {"max": max(steps_ms), "min": min(steps_ms), "distinct_devices": len({r["pci_bus_id"] for r in rank_info})}
        else:
            result["ranks"] = [{"rank": 0, "device": gpu, "device_count": ngpu_visible, "pci_bus_id": pci_bus_id(ctx), "shard_sequences": int(sum(c["nseq"] for c in chunks)),
                                "shard_residues": int(d_local), "chunks": len(chunks), "ms_per_step": round(elapsed_rank / args.steps * 1e3, 3)}]
        result["top_equals_single_gpu_reference_run"] = check_top_reference_run(args, nseq_total, strong, world, top)
        # not `value`: the same pass when the boundary hands over host buffers -- the reference's timed region
        # (FPGAsearch.c:80 -> :276: uploads + kernels + download of the score table).  The host buffers are pinned
        # (the reference allocates its own 64-byte aligned "for DMA", sequences.h:15, FPGAsearch.c:69-74), and the
        # upload of chunk k+1 is queued while chunk k is being searched (the library's upload stream), as the CLI does.
        result["inclusive"] = pcie_inclusive(ctx, chunks, nq, sum_m, d_local, pinned=True, steps=min(args.steps, 10))
        result["value_inclusive"] = result["inclusive"]["value"]
        result["inclusive"]["vs_resident"] = round(result["inclusive"]["ms_per_step"] / (elapsed / args.steps * 1e3) - 1.0, 4) if world == 1 else None
        result["pcie_inclusive"] = {"gcups": result["inclusive"]["value"], "ms": result["inclusive"]["ms_per_step"], "what": "= inclusive (the name of rounds 2 - 4)"}
        result["pcie_inclusive_pageable"] = pcie_inclusive(ctx, chunks, nq, sum_m, d_local, pinned=False, steps=2)["value"]
        if args.cpu_seconds > 0 and world == 1 and chunks:  # reported at N = 1 only
            result["cpu_baseline"] = cpu_baseline(args, a, m, a_disp, chunks, ctx, sm, wl, sum_m)
            # the metric's own clause, "top-10 score bit-exact", against the ORACLE (its scalar restatement) on the list the timed steps
            # produced: the nq x top listed pairs, every planted homolog (score in the downloaded tables, presence in the list)
            if strong:
                result["top10_oracle"] = top_oracle_pin(plan, shard, queries, sm, wl, top, chunks)
                result["top10_equals_oracle"] = result["top10_oracle"]["ok"]
    for c in chunks:
        ctx.chunk_release(c["h"])
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
        if result.get("top_equals_single_gpu_reference_run") is False:
            raise SystemExit("bench.py: the merged top list differs from the single-GPU run's list (tests/reference_runs/bench_top_*.json)")
        if result.get("top10_equals_oracle") is False:
            raise SystemExit("bench.py: the top list differs from the oracle: " + json.dumps(result["top10_oracle"]))


def pci_bus_id(ctx):
    for line in ctx.info().splitlines():
        if "PCI bus id" in line:
            return line.split(":", 1)[1].strip()
    return "unknown"


def pcie_inclusive(ctx, chunks, nq, sum_m, d_local, pinned, steps):
    """SURVEY 8(d)'s timed region on rank 0's chunks -- the reference's own (FPGAsearch.c:80 -> :276): H2D of the interleaved
    chunks + re-tile + search + D2H of the whole int32 score tables; uploads two chunks ahead of the search; a first chunk of
    some size is cut in two by the library at its upload (the device starts on the head while the rest is on the link).
    Measured like `value`: an untimed pass, then `steps` passes between waits for everything the device has, mean per pass.
    pinned: the host buffers are page-locked memory from the library (oswald_hip_host_alloc), allocated and filled once --
    a long-running caller allocates its DMA buffers once (the reference: posix_memalign, FPGAsearch.c:69-74), and the first
    DMA through a fresh page-locked allocation, like the first search into a fresh chunk slot, costs milliseconds."""
    from oswald_amd import capi
    keep, bufs, outs = [], [], []
    for c in chunks:
        if pinned:
            hb = [capi.pinned_copy(c[k]) for k in ("b", "n", "disp")] + [capi.HostBuffer((nq, len(c["n"]) * 16), np.int32)]
            keep += hb
            bufs.append((hb[0].a, hb[1].a, hb[2].a)); outs.append(hb[3].a)
        else:
            bufs.append((c["b"], c["n"], c["disp"])); outs.append(np.zeros((nq, len(c["n"]) * 16), np.int32))

    slot_bytes = max((b[0].size for b in bufs), default=0)
    slot_groups = max((len(b[1]) for b in bufs), default=0)

    def one_pass():
        # the device buffers of the chunk slots are created INSIDE the clock, like the reference's six clCreateBuffer behind its tick
        # (FPGAsearch.c:80, :85-96), and released outside it (:361-368, behind the tock at :276) -- round 6; until round 5 the first,
        # untimed pass made them for all the passes
        if slot_bytes:
            ctx.reserve_chunks(slot_bytes, slot_groups, 16, nq, min(len(bufs) + 1, 4))
        # two chunks ahead: the copies of chunk k+2 run beside the search of chunk k, its re-tile when that search drains, and
        # the host plans and queues the search of chunk k+1 meanwhile
        hs = [ctx.chunk_upload(*bufs[0], 16, wait=False)] if bufs else []
        for k in range(len(bufs)):
            ctx.chunk_search(hs[k], outs[k])             # waits for nothing; queued behind ITS upload and the search before it
            for j in ((1, 2) if k == 0 else (k + 2,)):
                if j < len(bufs):
                    hs.append(ctx.chunk_upload(*bufs[j], 16, wait=False))
            ctx.chunk_release(hs[k])                     # last: returns when the chunk's upload has landed; the slot is re-used once the device is through with it

    if slot_bytes:      # host staging of the slots: before the clock, like the reference's posix_memalign (FPGAsearch.c:69-74)
        ctx.reserve_host(slot_groups, 16, nq, min(len(bufs) + 1, 4))
    one_pass()          # untimed: first DMA through the host buffers
    ctx.wait()
    ctx.release_chunks()
    times = []
    for _ in range(max(1, steps)):
        t0 = time.perf_counter()
        one_pass()
        ctx.wait()      # (the tables have landed: a caller reads them now)
        times.append(time.perf_counter() - t0)
        ctx.release_chunks()   # outside the clock, like the reference's clReleaseMemObject
    t = sum(times) / len(times)
    pcie_inclusive.last_scores = [np.array(o) for o in outs] if pinned else outs   # (the pinned buffers go back to the library)
    for hb in keep:
        hb.close()
    return {"value": round(sum_m * d_local / t / 1e9, 1), "unit": "GCUPS", "ms_per_step": round(t * 1e3, 3), "steps": len(times), "ms_best": round(min(times) * 1e3, 3),
            "what": "SURVEY 8(d)'s timed region on rank 0's chunks (reference FPGAsearch.c:80-276): creation of the chunk slots' device buffers + H2D of the interleaved chunks + re-tile + search + D2H of "
                    "all int32 scores; uploads two chunks ahead of the search, a large first chunk cut in two at its upload; mean of the timed passes behind an untimed one; "
                    + ("page-locked host buffers (oswald_hip_host_alloc)" if pinned else "pageable host memory")}


def reference_run_path(args, nseq_total):
    return os.path.join(ROOT, "tests", "reference_runs", f"bench_top_{args.workload}_{nseq_total}.json")


def check_top_reference_run(args, nseq_total, strong, world, top):
    """The merged top-r list (scores and positions in the globally sorted database) against the list a
    single GPU produced for the same workload (written with --write-top-reference-run at N = 1; NOT an oracle
    result -- tests/test_reference_runs.py pins its scores to the oracle): True / False, or None when no such
    file exists for this workload (or in --weak mode, where the database differs)."""
    if not strong or top is None:
        return None
    sc, ix = top
    path = reference_run_path(args, nseq_total)
    if args.write_top_reference_run and world == 1:
        with open(path, "w") as f:
            json.dump({"workload": args.workload, "nseq": nseq_total, "top": args.top, "scores": sc.tolist(), "index": ix.tolist()}, f)
        return True
    try:
        with open(path) as f:
            g = json.load(f)
    except (OSError, ValueError):
        return None
    r = min(args.top, g["top"])
    return bool(np.array_equal(np.array(g["scores"])[:, :r], sc[:, :r]) and np.array_equal(np.array(g["index"])[:, :r], ix[:, :r]))


def top_oracle_pin(plan, shard, queries, sm, wl, top, chunks):
    """oracle/toppin.py on the list of the timed steps (N = 1): the scalar restatement on the listed (query, sequence) pairs and
    on the planted homologs, whose GPU scores come from the score tables the inclusive leg downloaded."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    import toppin
    tables = pcie_inclusive.last_scores
    starts = [(int(c["gpos"][0]), int(c["gpos"][-1]), k) for k, c in enumerate(chunks) if len(c["gpos"])]

    def score_at(q, pos):
        for lo, hi, k in starts:
            if lo <= pos <= hi:
                j = int(np.searchsorted(chunks[k]["gpos"], pos))
                if j < len(chunks[k]["gpos"]) and int(chunks[k]["gpos"][j]) == pos:
                    return tables[k][q, j]
        raise KeyError(pos)

    t0 = time.perf_counter()
    pin = toppin.pin_top_list(pyoracle, plan, shard.order, queries, sm, wl["go"], wl["ge"], top[0], top[1], score_at=score_at)
    pin["seconds"] = round(time.perf_counter() - t0, 2)
    pin["what"] = ("oracle/sw_oracle.c scalar restatement on the nq x top (query, listed sequence) pairs and on the planted homologs "
                   "(score in the downloaded table; a copy that outscores a list's last entry must be listed)")
    pin["first"] = [list(map(lambda v: None if v is None else int(v) if not isinstance(v, str) else v, x)) for x in pin["first"]]
    return pin


def measured_traffic(workload_name, nseq, dtype=None, launches_per_step=None):
    """HBM bytes per launch of the DP kernels from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    runs, see tools/profile_gpu.sh and DESIGN.md for the unit and the gfx950 correction) committed under
    profiles/ for this exact workload -- and for THIS kernel source: the summary carries a digest of the
    kernel sources it was measured on; a stale one is not reported."""
    if nseq is None:
        return None, "PMC passes are single-GPU runs"
    # a file per MODE (round 6): workload, database size, cell arithmetic, and whether the pairs' tails run (OSWALD_HIP_PAIR_TAILS, default 1)
    tails = os.environ.get("OSWALD_HIP_PAIR_TAILS", "1")
    path = os.path.join(ROOT, "profiles", f"traffic_{workload_name}_{nseq}" + ("" if tails == "1" else f"_tails{tails}") + ".json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC summary committed for this workload"
    if dtype is not None and t.get("dtype") not in (None, dtype):
        return None, f"the PMC summary committed for this workload was measured on the {t.get('dtype')} cells"
    if launches_per_step is not None and (t.get("launches_per_step") is None or abs(float(t["launches_per_step"]) - launches_per_step) > 0.01):
        return None, f"the PMC summary committed for this workload was measured at {t.get('launches_per_step')} launches per step (this run: {launches_per_step:g})"
    if t.get("pair_tails", "1") != tails:
        return None, f"the PMC summary committed for this workload was measured with OSWALD_HIP_PAIR_TAILS={t.get('pair_tails')}"
    if t.get("source_digest") != source_digest():
        print(f"bench.py: {path} was measured on other kernel sources (digest {t.get('source_digest')} != {source_digest()}); "
              "traffic not reported -- re-run tools/profile_gpu.sh", file=sys.stderr, flush=True)
        return None, "stale: PMC summary was measured on other kernel sources"
    return t.get("hbm_bytes_per_launch"), f"rocprofv3 FETCH_SIZE+WRITE_SIZE, {os.path.relpath(path, ROOT)}"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, a, m, a_disp, chunks, ctx, sm, wl, sum_m):
    """The oracle's SIMD port of the reference host path (SSE4.1/AVX2 int8->int16->int32, OpenMP over groups) timed on
    this box's host cores on a bounded sample of THE BENCHED DATABASE: every k-th W-lane group of every chunk of rank 0
    (the chunks are length-sorted runs, so the sample has the database's own length distribution), all queries.  Its
    scores are compared with the GPU's for the sampled sequences (the score tables the PCIe-inclusive leg downloaded)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from oswald_amd import dblayout
    W = args.cpu_lanes
    # the host threads this process may really keep busy: the box gives its container a CPU quota (16 CPUs' worth of a 256-thread host on
    # the round-5 pool) -- a team beyond it is throttled for most of every scheduling period, and `cores` would name threads that never ran
    from oswald_amd import hostinfo
    threads = max(1, min(pyoracle.max_threads(), hostinfo.usable_cpus()))
    gpu_tables = pcie_inclusive.last_scores

    def sample(stride):
        """every stride-th W-lane group of every chunk -> (per-chunk sequence picks, residues, interleaved sample)"""
        picks, lens_all, res_all = [], [], []
        for c in chunks:
            sl, sr, so = c["ls"], c["res"], c["off"]
            nseq = len(sl)
            gsel = np.arange(0, (nseq + W - 1) // W, stride)
            seqs = (gsel[:, None] * W + np.arange(W)[None, :]).reshape(-1)
            seqs = seqs[seqs < nseq]
            lens = sl[seqs].astype(np.int64)
            off = np.zeros(len(seqs) + 1, np.int64)
            np.cumsum(lens, out=off[1:])
            idx = np.repeat(so[seqs] - off[:-1], lens) + np.arange(int(off[-1]))
            picks.append(seqs); lens_all.append(lens); res_all.append(sr[idx])
        lens = np.concatenate(lens_all)
        order = np.argsort(lens, kind="stable")                  # the host path searches a length-sorted database
        res_cat = np.concatenate(res_all)
        off_cat = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(lens, out=off_cat[1:])
        sl2 = lens[order]
        off2 = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(sl2, out=off2[1:])
        idx = np.repeat(off_cat[:-1][order] - off2[:-1], sl2) + np.arange(int(off2[-1]))
        bb, nn, dd = dblayout.interleave(sl2, res_cat[idx], off2, W, round_to=1)
        return picks, order, int(lens.sum()), bb, nn, dd.astype(np.uint32)

    total_groups = sum((len(c["ls"]) + W - 1) // W for c in chunks)
    total_res = float(sum(int(c["off"][-1]) for c in chunks))
    picks, order, dres, bb, nn, dd = sample(max(1, total_groups // 64))     # calibrate on a few groups ...
    t0 = time.perf_counter()
    pyoracle.search_chunk_simd(a, m, a_disp, bb, nn, dd, W, sm, wl["go"], wl["ge"], 256, threads)
    rate = sum_m * dres / (time.perf_counter() - t0)
    want_res = rate * args.cpu_seconds / sum_m                             # ... then size the sample for ~cpu_seconds
    stride = max(1, int(np.ceil(total_res / max(want_res, 1.0))))
    picks, order, dres, bb, nn, dd = sample(stride)
    t0 = time.perf_counter()
    sc_cpu, stage = pyoracle.search_chunk_simd(a, m, a_disp, bb, nn, dd, W, sm, wl["go"], wl["ge"], 256, threads)
    t = time.perf_counter() - t0
    # parity of the sampled sequences against the GPU score tables
    gpu = np.concatenate([tab[:, p] for tab, p in zip(gpu_tables, picks)], axis=1)[:, order]
    equal = bool(np.array_equal(gpu, sc_cpu[:, :gpu.shape[1]]))
    nsamp = int(sum(len(p) for p in picks))
    return {"value": round(sum_m * dres / t / 1e9, 3), "unit": "GCUPS", "cores": threads, "cores_note": f"{threads} of {os.cpu_count()} hardware threads (what the cgroup's cpu.max / the affinity mask let this process keep busy)", "host_threads_visible": pyoracle.max_threads(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"every {stride}-th {W}-lane group of each of rank 0's {len(chunks)} chunks = of the whole benched database ({nsamp} sequences, {dres} residues, "
                      f"mean length {dres / max(nsamp, 1):.0f}) x all {len(m)} queries, {'AVX2' if W == 32 else 'SSE4.1'} int8->int16->int32 port, block 256, {t:.1f} s",
            "gpu_scores_equal_on_sample": equal,
            "cells_by_precision": {"int8": int(stage[0]), "int16": int(stage[1]), "int32": int(stage[2])}}


if __name__ == "__main__":
    main()
