#!/bin/bash
# tools/sweep_tail.sh -- how the phase-1 queue is eaten (both ends / heaviest first) and where the quads go, on the
# single-query workloads and on C2 (diag library: the knobs exist only there)
export OSWALD_HIP_USE_DIAG_LIB=1
run() { wl=$1; shift; echo -n "$wl $* : "; env "$@" python bench.py --workload $wl ${NSEQ:+--nseq $NSEQ} --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for wl in c5 q1 c2; do
  NSEQ=100000
  run $wl X=1
  run $wl OSWALD_HIP_ONE_ENDED_WG=1
  run $wl OSWALD_HIP_ONE_ENDED_WG=1 OSWALD_HIP_QUAD_FRAC=10
  run $wl OSWALD_HIP_QUAD_FRAC=10
  run $wl OSWALD_HIP_ONE_ENDED_WG=1 OSWALD_HIP_QUAD_FRAC=0.25
  run $wl OSWALD_HIP_ONE_ENDED_WG=1 OSWALD_HIP_NO_PRIO=1
  run $wl OSWALD_HIP_ONE_ENDED_WG=1 OSWALD_HIP_TARGET_DIV=2
  run $wl OSWALD_HIP_ONE_ENDED_WG=1 OSWALD_HIP_TARGET_DIV=4
done
