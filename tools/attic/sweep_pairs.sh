#!/bin/bash
export OSWALD_HIP_USE_DIAG_LIB=1
run() { n=$1; shift; echo -n "nseq $n $* : "; for i in 1 2; do env "$@" python bench.py --nseq $n --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], end='  ')"; done; echo; }
for n in 100000 1000000; do
  run $n X=1
  run $n OSWALD_HIP_PAIRS=2
  run $n OSWALD_HIP_PAIR_MARGIN=0.95
  run $n OSWALD_HIP_PAIR_MARGIN=0.9
  run $n OSWALD_HIP_PAIR_MARGIN=0.8
done
