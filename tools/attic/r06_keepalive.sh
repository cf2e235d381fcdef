#!/bin/bash
# round 6 experiment: does a small kernel that stays busy beside the searches keep the clock up in SHORT launches (Q1: 2.1-2.2 GHz inside the kernel)?
run() { python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'incl', d['inclusive']['value'])"; }
for g in none 8 64; do
  if [ $g = none ]; then export OSWALD_HIP_WARM_MS=250; unset OSWALD_HIP_WARM_KEEP OSWALD_HIP_WARM_GRID; else export OSWALD_HIP_WARM_KEEP=1 OSWALD_HIP_WARM_MS=15000 OSWALD_HIP_WARM_GRID=$g; fi
  echo "== keep-alive workgroups: $g"
  echo "  q1 100k"; run --workload q1 --steps 300 --warmup 20
  echo "  q1 1m";   run --workload q1 --nseq 1000000 --steps 40 --warmup 5
  echo "  c5 100k"; run --workload c5 --steps 40 --warmup 5
  echo "  c2 100k"; run --nseq 100000 --steps 30 --warmup 5
  echo "  q1 1m, clock inside the kernel (diag library)"; OSWALD_HIP_USE_DIAG_LIB=1 OSWALD_HIP_DEBUG_TIMES=1 python bench.py --workload q1 --nseq 1000000 --steps 2 --warmup 1 --cpu-seconds 0 2>&1 >/dev/null | grep "core clock" | tail -2
done
