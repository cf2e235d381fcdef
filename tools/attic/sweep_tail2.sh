#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; shift; echo -n "$wl $* : "; env "$@" python bench.py --workload $wl --nseq 100000 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for wl in c5 q1 c2; do
  run $wl X=1
  run $wl OSWALD_HIP_TARGET_DIV=1.0
  run $wl OSWALD_HIP_TARGET_DIV=0.8
  run $wl OSWALD_HIP_QUAD_FRAC=0.35
  run $wl OSWALD_HIP_COL_COST=20
  run $wl OSWALD_HIP_WG_MINCOLS=1024
  run $wl OSWALD_HIP_WG_MINCOLS=4096
  run $wl OSWALD_HIP_TWO_ENDED=1
done
export OSWALD_HIP_DEBUG_TIMES=1
for wl in c5 q1; do python bench.py --workload $wl --steps 2 --warmup 1 --cpu-seconds 0 2>&1 >/dev/null | grep -m 6 "CUs;\|DP launch span"; done
