#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; shift; echo -n "$wl $* : "; env "$@" python bench.py --workload $wl --nseq 100000 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], round(d['planned_spill_bytes_per_step']/1e9,2))"; }
for wl in c2 c5 q1; do
  run $wl X=1
  run $wl OSWALD_HIP_TARGET_DIV=1.0
  run $wl OSWALD_HIP_TARGET_DIV=1.6
  run $wl OSWALD_HIP_QUAD_FRAC=0.35
  run $wl OSWALD_HIP_QUAD_FRAC=0.7
  run $wl OSWALD_HIP_WG_MINCOLS=1024
  run $wl OSWALD_HIP_WG_MINCOLS=3000
  run $wl OSWALD_HIP_PAIR_MARGIN=1.0
  run $wl OSWALD_HIP_PAIR_MARGIN=1.08
done
