#!/usr/bin/env python3
"""bench.py -- GCUPS of the MI355X Smith-Waterman search path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every query of the
workload against the rank's resident database shard (DP kernels incl. the
exact int32 re-run of saturated cells, on-device top-r) and, for N > 1, the
gather of the per-GPU top-r lists over RCCL.  Inputs (re-tiled residues,
query profile) are resident in HBM before the timed region starts.

Workload at N = 1: BASELINE.json configs[1] -- 20 queries of length 100..1000
(sum 11 000) against a 100k-sequence synthetic length-binned database
(~36.5 M residues), BLOSUM62, gap 10/2, int16 cells (packed, exact below 22256;
sequences above are re-run in int32).  For N > 1 every
rank holds its own 100k-sequence shard (weak scaling; shards are chunk-sharded
parts of an N x 100k-sequence database, no data-path collective).

GCUPS = sum(query lengths) x unpadded database residues / seconds / 1e9, the
reference's definition (reference host/src/FPGAsearch.c:324).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9                 # max clock (MI355X_MICROARCH.md)
N_CU = 256
SIMD_PER_CU = 4
PK_ISSUE_CYCLES = 4.0            # packed 16-bit VOP3P: one wave instruction per 4 cycles per SIMD (tools/ubench.hip, measured 4.2-4.5)
# VALU instructions per wave per query row (= per 128 cells) of the DP kernels: {first-pass arithmetic: (one query per
# lane: two sequences per lane, incl. the v_perm_b32 that pairs their scores; query pairs)}
PK_OPS_PER_ROW = {16: (7.5, 6.5), 32: (24.0, 24.0)}
DTYPE = {16: "int16", 32: "int32"}
CELL_LABEL = {16: "int16 cells (packed, column frames, exact < 22256), int32 re-run above", 32: "int32 cells"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=100000, help="database sequences per GPU")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "q1"])
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--cell-bits", type=int, default=16, choices=[16, 32],
                    help="cell arithmetic: 16 = packed int16 (the cells BASELINE.json names; the library's default), 32 = int32 only")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 = skip)")
    ap.add_argument("--cpu-lanes", type=int, default=32, choices=[16, 32], help="16 = SSE4.1 port, 32 = AVX2 port")
    return ap.parse_args()


def workload(name):
    if name == "c2":
        return dict(qlens=None, matrix="blosum62", go=10, ge=2, label="C2: 20 queries len 100-1000 x {nseq}-seq synthetic DB per GPU, BLOSUM62 10/2")
    if name == "c3":
        return dict(qlens=None, matrix="pam250", go=14, ge=2, label="C3: 20 queries len 100-1000 x {nseq}-seq synthetic DB per GPU, PAM250 14/2")
    if name == "q1":
        return dict(qlens=[375], matrix="blosum62", go=10, ge=2, label="Q1: 1 query len 375 (the C1 query) x {nseq}-seq synthetic DB per GPU, BLOSUM62 10/2")
    return dict(qlens=[5000], matrix="blosum62", go=10, ge=2, label="C5: 1 query len 5000 x {nseq}-seq synthetic DB per GPU, BLOSUM62 10/2")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the search path is HIP only (no CPU fallback)")
    # one rank per GPU over RCCL ("nccl").  OSWALD_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer
    # GPUs than ranks: ranks share the visible GPUs and the top-r gather runs over gloo on host tensors.
    backend = os.environ.get("OSWALD_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    dist = None
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from oswald_amd import capi, dblayout, multigpu, submat, synth

    wl = workload(args.workload)
    qlens = wl["qlens"] or synth.default_query_lengths()
    queries = synth.make_queries(qlens)
    sm = submat.load(wl["matrix"])
    m = np.array(qlens, dtype=np.uint16)
    a_disp = np.concatenate([[0], np.cumsum(m[:-1], dtype=np.int64)]).astype(np.uint32)
    a = np.concatenate(queries)
    sum_m = int(m.astype(np.int64).sum())

    # this rank's shard: generated, sorted by length, interleaved exactly like the
    # reference's preprocessed + assembled database (W = 16, pad to x28)
    t0 = time.time()
    L, R, O = synth.make_database(args.nseq, queries, seed=synth.SEED_DB + 1000003 * rank)
    order, sl, sr, so = dblayout.sort_by_length(L, R, O)
    b, n, disp = dblayout.interleave(sl, sr, so, 16)
    d_local = int(sl.astype(np.int64).sum())
    t_gen = time.time() - t0

    ctx = capi.Context(1, [gpu])
    cell_bits = args.cell_bits
    ctx.set_scoring(sm, wl["go"], wl["ge"], cell_bits)
    ctx.set_queries(a, m, a_disp)
    if b.size >= 2**32:
        raise SystemExit("shard too large for one chunk")
    chunk = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
    geom = ctx.chunk_geometry(chunk)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        ctx.chunk_search(chunk, None)
        sc, ix = ctx.chunk_topr(chunk, args.nseq, args.top)   # syncs the library's stream
        gix = ix.astype(np.int64) + rank * args.nseq          # global index = shard base + sorted position
        return multigpu.gather_topr(sc, gix, args.top, dist, coll_dev if dist is not None else None)

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
    ctx.set_profiling(False)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    d_all = torch.tensor([d_local], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(d_all, op=dist.ReduceOp.SUM)
    elapsed = float(t_all.item())
    d_total = float(d_all.item())

    result = None
    if rank == 0:
        cells_per_step = sum_m * d_total
        gcups = cells_per_step * args.steps / elapsed / 1e9
        # roofline of the dominant kernel (osw_sw_pk16 + its int32 re-run, one event pair per step)
        nq = len(qlens)
        alg_bytes = nq * (geom["residue_bytes_per_query"] + 4 * args.nseq) + sum_m + 768
        kern_s = kern_ms / max(1, kern_launches) / 1e3
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        kern_gcups = sum_m * d_local / kern_s / 1e9 if kern_s > 0 else 0.0
        ops_row = PK_OPS_PER_ROW[cell_bits][1 if nq > 1 else 0]  # a multi-query search runs (mostly) as query pairs
        valu_ceiling = N_CU * SIMD_PER_CU * (CLOCK_HZ / PK_ISSUE_CYCLES) * 128.0 / ops_row / 1e9
        kname = {16: "osw_sw_s16q+osw_sw_s16(+osw_sw_i32)", 32: "osw_sw_i32"}[cell_bits]
        traffic = measured_traffic(args.workload, args.nseq)
        result = {
            "metric": "GCUPS", "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[cell_bits], "data": "synthetic",
            "config": {"workload": wl["label"].format(nseq=args.nseq) + ", " + CELL_LABEL[cell_bits], "queries": nq, "query_residues": sum_m, "db_sequences_per_gpu": args.nseq,
                       "db_residues_total": int(d_total), "matrix": wl["matrix"], "gap_open": wl["go"], "gap_extend": wl["ge"],
                       "top": args.top, "sharding": f"db-shard x{world}, {'RCCL' if backend == 'nccl' else backend} all_gather of top-{args.top}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": kname, "kernel_ms": round(kern_s * 1e3, 3), "kernel_gcups": round(kern_gcups, 1),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "valu": {"ceiling_gcups": round(valu_ceiling, 0), "frac": round(kern_gcups / valu_ceiling, 4),
                                  "instr_per_128_cells": ops_row,
                                  "note": "the DP is VALU-issue bound: instr_per_128_cells VALU instructions per wave per query row (query-pair cell for a multi-query search), one packed 16-bit instruction per 4 cycles per SIMD"}},
            # the north star's ">= 0.5 x HBM roofline" is only well posed under the reference's own traffic
            # model, 1 B of substitution score per cell streamed from device DRAM (sw.cl:57): 8 TB/s = 8000 GCUPS
            "reference_traffic_model": {"bytes_per_cell": 1.0, "roofline_gcups": HBM_PEAK_GBS, "frac": round(gcups / world / HBM_PEAK_GBS, 4)},
            "rerun_items_int32": int(rerun), "work_items": int(ctx.chunk_geometry(chunk)["work_items"]), "max_log2_geometry": int(ctx.chunk_geometry(chunk)["max_log2_geometry"]), "top1_scores": [int(x) for x in top[0][:, 0]] if top is not None else None,
            "setup_s": round(t_gen, 1),
        }
        # not `value`: the same pass when the boundary hands over host buffers (H2D of the interleaved
        # chunk + re-tile + search + D2H of the full int32 score table), the reference's timed region
        out_full = np.zeros((nq, len(n) * 16), np.int32)
        t0 = time.perf_counter()  # (rank 0's shard; the other ranks idle at the final barrier meanwhile)
        h2 = ctx.chunk_upload(b, n, disp.astype(np.uint32), 16)
        ctx.chunk_search(h2, out_full)
        ctx.wait()
        t_pcie = time.perf_counter() - t0
        ctx.chunk_release(h2)
        result["pcie_inclusive"] = {"gcups": round(sum_m * d_local / t_pcie / 1e9, 1), "ms": round(t_pcie * 1e3, 2),
                                    "what": "chunk_upload (H2D + re-tile) + search + D2H of all scores, pageable host memory"}
        if args.cpu_seconds > 0 and world == 1:  # reported at N = 1 only
            result["cpu_baseline"] = cpu_baseline(args, a, m, a_disp, sl, sr, so, sm, wl, sum_m, ctx, chunk, n)
    ctx.chunk_release(chunk)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def measured_traffic(workload_name, nseq):
    """HBM bytes per launch of the DP kernel from rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, separate runs, see tools/profile_gpu.sh and
    DESIGN.md for the unit and the gfx950 correction), if a summary for this
    exact workload has been committed under profiles/; else None."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload_name}_{nseq}.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(args, a, m, a_disp, sl, sr, so, sm, wl, sum_m, ctx, chunk, n_gpu):
    """The oracle's SIMD port of the reference host path (SSE4.1/AVX2
    int8->int16->int32, OpenMP over groups) timed on this box's host cores on a
    bounded sample: every k-th W-lane group of the sorted shard, all queries.
    Its scores are also compared with the GPU's for the sampled sequences."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from oswald_amd import dblayout
    W = args.cpu_lanes
    threads = pyoracle.max_threads()
    nseq = len(sl)
    ngroups = (nseq + W - 1) // W
    # calibrate on a few groups, then size the sample for ~cpu_seconds
    def sample(stride):
        gsel = np.arange(0, ngroups, stride)
        seqs = (gsel[:, None] * W + np.arange(W)[None, :]).reshape(-1)
        seqs = seqs[seqs < nseq]
        lens = sl[seqs].astype(np.int64)
        off = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        idx = np.repeat(so[seqs] - off[:-1], lens) + np.arange(int(off[-1]))
        res = sr[idx]
        bb, nn, dd = dblayout.interleave(lens.astype(np.uint16), res, off, W, round_to=1)
        return seqs, int(lens.sum()), bb, nn, dd.astype(np.uint32)
    seqs, dres, bb, nn, dd = sample(max(1, ngroups // 64))
    t0 = time.perf_counter()
    pyoracle.search_chunk_simd(a, m, a_disp, bb, nn, dd, W, sm, wl["go"], wl["ge"], 256, threads)
    t_cal = time.perf_counter() - t0
    rate = sum_m * dres / t_cal
    want_res = rate * args.cpu_seconds / sum_m
    stride = max(1, int(np.ceil(float(sl.astype(np.int64).sum()) / max(want_res, 1.0))))
    seqs, dres, bb, nn, dd = sample(stride)
    t0 = time.perf_counter()
    sc_cpu, stage = pyoracle.search_chunk_simd(a, m, a_disp, bb, nn, dd, W, sm, wl["go"], wl["ge"], 256, threads)
    t = time.perf_counter() - t0
    # parity of the sampled sequences against the GPU score table
    out = np.zeros((len(m), len(n_gpu) * 16), np.int32)
    ctx.chunk_search(chunk, out)
    ctx.wait()
    equal = bool(np.array_equal(out[:, seqs], sc_cpu[:, :len(seqs)]))
    return {"value": round(sum_m * dres / t / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
            "sample": f"every {stride}-th {W}-lane group of the sorted shard ({len(seqs)} sequences, {dres} residues) x all {len(m)} queries, "
                      f"{'AVX2' if W == 32 else 'SSE4.1'} int8->int16->int32 port, block 256, {t:.1f} s",
            "gpu_scores_equal_on_sample": equal,
            "cells_by_precision": {"int8": int(stage[0]), "int16": int(stage[1]), "int32": int(stage[2])}}


if __name__ == "__main__":
    main()
