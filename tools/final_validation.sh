#!/bin/bash
# tools/final_validation.sh -- the round's evidence in one GPU call: the -m gpu suite, the bench lines the documents
# quote, the multi-rank rehearsal, the shard-balance prediction, rocprofv3 summaries.  Everything lands under
# gpurun_out/final/ (scratch); tools/collect_final.py copies what is cited into profiles/.
OUT=gpurun_out/final
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
b() { name=$1; shift; python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d.get('top_equals_single_gpu_golden'))" 2>/dev/null)"; }
b c4_1gpu --steps 20 --warmup 5
b c2 --nseq 100000 --steps 20 --warmup 5
b c3_int8 --workload c3 --steps 10 --warmup 2
b c3_int16 --workload c3 --cell-bits 16 --steps 20 --warmup 5 --cpu-seconds 0
b c5 --workload c5 --steps 20 --warmup 5
b q1 --workload q1 --steps 50 --warmup 10
OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29641 python bench.py --gpus 4 --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; echo "gloo4 rc=$?"
OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29642 python bench.py --gpus 4 --shard-rule reference --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_gloo4_reference_rule.json 2> $OUT/bench_gloo4_reference_rule.err; echo "gloo4 reference rule rc=$?"
python tests/shard_balance_gpu.py 2 4 8 > $OUT/shard_balance.txt 2>&1; echo "shard balance rc=$?"; cat $OUT/shard_balance.txt
for spec in "c2 1000000" "c3 100000" "c5 100000"; do bash tools/profile_gpu.sh $spec > $OUT/prof_$(echo $spec | tr ' ' '_').log 2>&1; echo "profile $spec rc=$?"; done
timeout -k 10 200 ./tools/oprate_q8 > $OUT/oprate_q8.txt 2>&1
python tools/cli_e2e.py 1000000 > $OUT/cli_1m.txt 2>&1; echo "cli e2e rc=$?"; tail -5 $OUT/cli_1m.txt
