#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; shift; echo -n "$wl $* : "; env "$@" python bench.py --workload $wl --nseq 100000 --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['work_items'])"; }
for wl in c5 q1; do
  run $wl X=1
  run $wl X=2
  run $wl OSWALD_HIP_WG_MINCOLS=512
  run $wl OSWALD_HIP_WG_MINCOLS=256
  run $wl OSWALD_HIP_WG_MINCOLS=0
  run $wl OSWALD_HIP_WG_MINCOLS=0 OSWALD_HIP_TARGET_DIV=2
  run $wl OSWALD_HIP_WG_MINCOLS=256 OSWALD_HIP_TARGET_DIV=2
  run $wl OSWALD_HIP_WG_MINCOLS=0 OSWALD_HIP_WG_WIDECOLS=0
done
