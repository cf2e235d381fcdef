#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { n=$1; shift; echo -n "nseq $n $* : "; env "$@" python bench.py --nseq $n --steps 6 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], round(d['planned_spill_bytes_per_step']/1e9,2))"; }
for n in 100000 1000000; do
  for m in 1.03 1.08 1.12 1.2; do run $n OSWALD_HIP_PAIR_MARGIN=$m; done
  run $n OSWALD_HIP_PAIR_MARGIN=1.08 OSWALD_HIP_TARGET_DIV=1.0
  run $n OSWALD_HIP_PAIR_MARGIN=1.12 OSWALD_HIP_TARGET_DIV=1.0
done
