#!/bin/bash
# tools/sweep_many.sh -- planner knobs, one at a time, on the single-query workloads (diagnostic)
WLS=${1:-"c5 q1"}
run() { # label, env assignments...
  label=$1; shift
  for wl in $WLS; do
    r=$(env "$@" python bench.py --workload $wl --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'])")
    echo "$wl $label -> $r"
  done
}
run default X=1
run FORCE_WG=0 OSWALD_HIP_FORCE_WG=0
run FORCE_WG=1 OSWALD_HIP_FORCE_WG=1
run NO_PRIO OSWALD_HIP_NO_PRIO=1
run TWO_ENDED OSWALD_HIP_TWO_ENDED=1
for v in 0 512 1024 4096 100000; do run WG_MINCOLS=$v OSWALD_HIP_WG_MINCOLS=$v; done
for v in 0 1024 8192; do run WG_WIDECOLS=$v OSWALD_HIP_WG_WIDECOLS=$v; done
for v in 2 20 40; do run COL_COST=$v OSWALD_HIP_COL_COST=$v; done
for v in 0.5 3 6; do run TARGET_DIV=$v OSWALD_HIP_TARGET_DIV=$v; done
for lg in 1 2 3 4 5; do run "FORCE_LG=$lg,WG=0" OSWALD_HIP_FORCE_LG=$lg OSWALD_HIP_FORCE_WG=0; done
for lg in 2 3 4 5 6; do run "FORCE_LG=$lg,WG=1" OSWALD_HIP_FORCE_LG=$lg OSWALD_HIP_FORCE_WG=1; done
