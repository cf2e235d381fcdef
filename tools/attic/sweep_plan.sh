#!/bin/bash
# tools/sweep_plan.sh -- planner threshold sweep on the GPU box (diagnostic): prints GCUPS per setting.
# usage: sweep_plan.sh VAR "v1 v2 ..." [workloads]
VAR=${1:-OSWALD_HIP_TARGET_DIV}; VALS=${2:-"1.5 2 3 4"}; WLS=${3:-"c2 c5 q1"}
for wl in $WLS; do
  for v in $VALS; do
    r=$(env $VAR=$v python bench.py --workload $wl --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
    echo "$wl $VAR=$v -> $r"
  done
done
