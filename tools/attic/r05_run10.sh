#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/t5.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/t5.txt
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;l=[x for x in open('$O/bench_$name.json').read().split(chr(10)) if x.startswith('{')];d=json.loads(l[-1]);print(d['value'],d['ms_per_step'],'kernel',d['roofline']['kernel_ms'],d['roofline']['kernel_gcups'],'incl',d['inclusive']['value'],d.get('top_equals_single_gpu_golden'))" 2>/dev/null)"; }
b q1_100k --workload q1 --steps 50 --warmup 10 --cpu-seconds 0
b q1_1m --workload q1 --nseq 1000000 --steps 20 --warmup 5 --cpu-seconds 0
b c5 --workload c5 --steps 20 --warmup 5 --cpu-seconds 0
b c5_1m --workload c5 --nseq 1000000 --steps 5 --warmup 1 --cpu-seconds 0
b c2 --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0
b c4_1gpu --steps 10 --warmup 3 --cpu-seconds 0
b c3_int8 --workload c3 --steps 10 --warmup 2 --cpu-seconds 0
