#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; n=$2; shift; shift; echo -n "$wl $n $* : "; env "$@" python bench.py --workload $wl --nseq $n --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['work_items'], d['top_equals_single_gpu_golden'])"; }
run c5 100000 X=1
run q1 100000 X=1
run c2 100000 X=1
run c2 1000000 X=1
run c3 100000 X=1
run c5 100000 OSWALD_HIP_TARGET_DIV=1.6
run q1 100000 OSWALD_HIP_TARGET_DIV=1.6
run c5 100000 OSWALD_HIP_TARGET_DIV=2.5
run q1 100000 OSWALD_HIP_TARGET_DIV=2.5
run c5 100000 OSWALD_HIP_QUAD_FRAC=0.25
run q1 100000 OSWALD_HIP_QUAD_FRAC=0.25
