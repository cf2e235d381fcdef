#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/t6.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/t6.txt
b() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;l=[x for x in open('$O/bench_$name.json').read().split(chr(10)) if x.startswith('{')];d=json.loads(l[-1]);print(d['value'],d['ms_per_step'],'kernel',d['roofline']['kernel_ms'],d['roofline']['kernel_gcups'],'rerun',d['rerun_ms_per_step'],d['rerun_items_int32'],d.get('top_equals_single_gpu_golden'), (d.get('cpu_baseline') or {}).get('gpu_scores_equal_on_sample'))" 2>/dev/null)"; }
b c2_int32 --nseq 100000 --cell-bits 32 --steps 3 --warmup 1 --cpu-seconds 0
b hi --workload hi --steps 5 --warmup 1 --cpu-seconds 8
