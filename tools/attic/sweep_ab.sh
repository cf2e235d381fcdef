#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { n=$1; shift; echo -n "nseq $n $* : "; for i in 1 2 3; do env "$@" python bench.py --nseq $n --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], end=' ')"; done; echo; }
for n in 100000 1000000; do
  run $n X=1
  run $n OSWALD_HIP_TARGET_DIV=1.0
  run $n OSWALD_HIP_WG_MINCOLS=3000
  run $n OSWALD_HIP_TARGET_DIV=1.0 OSWALD_HIP_WG_MINCOLS=3000
  run $n OSWALD_HIP_QUAD_FRAC=0.35
done
