#!/bin/bash
# tools/rebench.sh -- the bench lines once more, after tools/profile_gpu.sh has written the traffic files of the same
# sources (so that every line carries roofline.traffic)
OUT=gpurun_out/final
mkdir -p $OUT
b() { name=$1; shift; python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;l=[x for x in open('$OUT/bench_$name.json').read().split(chr(10)) if x.startswith('{')];d=json.loads(l[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['traffic'])" 2>/dev/null)"; }
b c4_1gpu --steps 20 --warmup 5
b c2 --nseq 100000 --steps 20 --warmup 5
b c3_int8 --workload c3 --steps 10 --warmup 2
b c5 --workload c5 --steps 20 --warmup 5
b q1 --workload q1 --steps 50 --warmup 10
