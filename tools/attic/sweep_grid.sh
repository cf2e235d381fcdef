#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; shift; echo -n "$wl $* : "; env "$@" python bench.py --workload $wl --nseq 100000 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for wl in c5 q1 c2; do for g in 4 3 2; do run $wl OSWALD_HIP_GRID_PER_CU=$g; done; done
