"""bench.py contract (GPU): one JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema(first_pass):
    if first_pass != "i16":
        pytest.skip("one run is enough: bench.py picks its own first-pass mode")
    env = dict(os.environ)
    env.pop("OSWALD_HIP_CELL_BITS", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--nseq", "6000",
                        "--cpu-seconds", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "GCUPS" and d["unit"] == "GCUPS" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "resident in HBM" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["gpu_scores_equal_on_sample"] is True
    assert d["value"] > 100 and d["dtype"] == "int16"
    # the reference's own timed region (upload + kernels + download), measured in the same run over its own timed steps
    i = d["inclusive"]
    assert i["unit"] == "GCUPS" and i["steps"] == 2 and i["value"] == d["value_inclusive"] and i["value"] > 0 and "FPGAsearch.c:80-276" in i["what"]   # (at 6 000 sequences either region may be the faster one)
    assert len(d["ranks"]) == 1 and d["ranks"][0]["pci_bus_id"] != "unknown" and d["ranks"][0]["device_count"] >= 1


def _bench(args, env):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_sharded_ranks_rehearsal(first_pass):
    """`bench.py --gpus 2` started bare: it spawns its two ranks (one process per rank, here sharing the box's one
    GPU, top-r gather over gloo instead of RCCL), shards ONE database (by dealt wave blocks, and by the reference's
    chunk rule) and reports strong scaling; the merged top-1 scores are those of the single-GPU run on the same database."""
    if first_pass != "i16":
        pytest.skip("one run is enough")
    env = dict(os.environ)
    env.pop("OSWALD_HIP_CELL_BITS", None)
    common = ["--steps", "2", "--warmup", "1", "--nseq", "40000", "--cpu-seconds", "0", "--max-chunk", "4000000"]
    one = _bench(["--gpus", "1"] + common, env)
    two = _bench(["--gpus", "2"] + common, dict(env, OSWALD_BENCH_BACKEND="gloo", MASTER_PORT="29611"))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "sharded over 2 GPUs" in two["config"]["sharding"]
    assert one["config"]["chunks_rank0"] >= 3 and two["config"]["chunks_rank0"] >= 2          # several chunks, dealt round-robin
    assert two["config"]["db_residues_total"] == one["config"]["db_residues_total"]            # one database, not one per rank
    assert two["top1_scores"] == one["top1_scores"] and two["value"] > 100 and two["config"]["shard_rule"] == "deal"
    # every rank says what it ran on and how long its steps took (the first line from a real multi-GPU node explains itself)
    assert [r["rank"] for r in two["ranks"]] == [0, 1] and all(r["pci_bus_id"] != "unknown" and r["device_count"] >= 1 and r["shard_residues"] > 0 and r["ms_per_step"] > 0 for r in two["ranks"])
    assert sum(r["shard_residues"] for r in two["ranks"]) == two["config"]["db_residues_total"] and two["rank_ms_per_step"]["max"] >= two["rank_ms_per_step"]["min"] > 0
    ref = _bench(["--gpus", "2", "--shard-rule", "reference"] + common, dict(env, OSWALD_BENCH_BACKEND="gloo", MASTER_PORT="29612"))
    assert ref["top1_scores"] == one["top1_scores"] and ref["config"]["db_residues_total"] == one["config"]["db_residues_total"]
    assert "chunk rule" in ref["config"]["sharding"] and ref["config"]["chunks_rank0"] >= 2


def _reference_run_top(nseq):
    with open(os.path.join(ROOT, "tests", "reference_runs", f"bench_top_c2_{nseq}.json")) as f:
        return json.load(f)


def test_bench_library_gather_at_world_size_one(first_pass):
    """`bench.py --gpus 1 --comm`: the one rank joins a process-level RCCL communicator made through the C ABI, and every
    step's oswald_hip_topr runs the all-gather + fold of N > 1 (at world size 1).  The line says who carried the gather
    and how many ranks RCCL itself reported; the merged top-10 is the committed single-GPU run's list (pinned to the oracle by tests/test_reference_runs.py)."""
    if first_pass != "i16":
        pytest.skip("one run is enough")
    env = dict(os.environ)
    env.pop("OSWALD_HIP_CELL_BITS", None)
    d = _bench(["--gpus", "1", "--comm", "--steps", "2", "--warmup", "1", "--nseq", "100000", "--cpu-seconds", "0"], env)
    c = d["config"]
    assert c["collective_backend"] == "RCCL (nccl)" and c["collective_note"] is None and c["collective_ranks"] == 1
    assert "liboswald_hip.so" in c["collective_via"] and "ncclAllGather" in c["collective_via"]
    assert d["top_equals_single_gpu_reference_run"] is True
    assert d["top1_scores"] == [row[0] for row in _reference_run_top(100000)["scores"]]


def test_bench_two_gpus_over_rccl(first_pass):
    """The N > 1 path on real hardware, wherever the box has two GPUs (the round's own test box has one: skipped
    there, runs by itself on a multi-GPU node): `bench.py --gpus 2`, one rank per GPU, RCCL -- no rehearsal backend,
    no fallback -- with the gather inside the C ABI and, second run, through torch.distributed.  RCCL must have seen
    two ranks, and the merged top-10 must be the committed single-GPU run's list under both shard rules."""
    if first_pass != "i16":
        pytest.skip("one run is enough")
    from oswald_amd import capi
    if capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % capi.device_count())
    env = dict(os.environ)
    env.pop("OSWALD_HIP_CELL_BITS", None)
    env.pop("OSWALD_BENCH_BACKEND", None)
    common = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--nseq", "100000", "--cpu-seconds", "0"]
    gold = [row[0] for row in _reference_run_top(100000)["scores"]]
    for k, extra in enumerate((["--gather", "lib"], ["--gather", "torch"], ["--gather", "lib", "--shard-rule", "reference"])):
        d = _bench(common + extra, dict(env, MASTER_PORT=str(29621 + k)))
        c = d["config"]
        assert d["n_gpus"] == 2 and d["scaling"] == "strong"
        assert c["collective_backend"] == "RCCL (nccl)" and c["collective_note"] is None and c["collective_ranks"] == 2, c
        assert ("liboswald_hip.so" in c["collective_via"]) == (extra[1] == "lib")
        assert d["top_equals_single_gpu_reference_run"] is True and d["top1_scores"] == gold
        assert d["value"] > 100
    # the library's communicator cannot be made (test hook: no rank joins it): the ranks decide TOGETHER to carry their lists
    # through torch.distributed -- still RCCL, and the line says so; asked for by name, `--gather lib` ends non-zero instead
    d = _bench(common, dict(env, MASTER_PORT="29626", OSWALD_BENCH_FAIL_LIB_COMM="1"))
    c = d["config"]
    assert c["collective_backend"] == "RCCL (nccl)" and c["collective_ranks"] == 2 and c["collective_via"] == "torch.distributed.all_gather"
    assert "oswald_hip_comm_init_rank failed" in c["collective_note"]
    assert d["top_equals_single_gpu_reference_run"] is True and d["top1_scores"] == gold
    # ... and the partial case: ONE rank reports a failure, the other holds a working communicator -- both must give it up
    # (oswald_hip_comm_destroy) before the lists go through torch, or the rank that kept it would all-gather alone inside oswald_hip_topr
    d = _bench(common, dict(env, MASTER_PORT="29628", OSWALD_BENCH_FAIL_LIB_COMM="rank:1"))
    c = d["config"]
    assert c["collective_ranks"] == 2 and c["collective_via"] == "torch.distributed.all_gather" and "another rank" in c["collective_note"]
    assert d["top_equals_single_gpu_reference_run"] is True and d["top1_scores"] == gold
    assert len(d["ranks"]) == 2 and d["rank_ms_per_step"]["distinct_devices"] == 2 and all(r["device_count"] >= 2 for r in d["ranks"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--gather", "lib"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env, MASTER_PORT="29627", OSWALD_BENCH_FAIL_LIB_COMM="1"))
    assert r.returncode != 0 and "communicator could not be made" in r.stderr
