#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; n=$2; shift; shift; echo -n "$wl $n $* : "; env "$@" python bench.py --workload $wl --nseq $n --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['work_items'])"; }
for wl in q1 c5; do
run $wl 100000 X=1
run $wl 100000 OSWALD_HIP_TARGET_DIV=1.0
run $wl 100000 OSWALD_HIP_TARGET_DIV=0.7
run $wl 100000 OSWALD_HIP_TARGET_DIV=0.5
run $wl 100000 OSWALD_HIP_TARGET_DIV=0.35
run $wl 100000 OSWALD_HIP_NO_PRIO=1
done
