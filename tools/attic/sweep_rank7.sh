#!/bin/bash
# tools/sweep_rank7.sh -- planner knobs on the last (longest-sequence) shard of the C4 database at 8 ranks (diagnostic)
run() { label=$1; shift; r=$(env ONLY_RANK=7 "$@" python tests/shard_balance_gpu.py 8 2>&1 | grep -o "shard ms \[[0-9.]*\]"); echo "$label -> $r"; }
run default X=1
for v in 512 1024 4096 100000; do run WG_MINCOLS=$v OSWALD_HIP_WG_MINCOLS=$v; done
for v in 0.75 1 2 3; do run TARGET_DIV=$v OSWALD_HIP_TARGET_DIV=$v; done
for v in 0.25 1.0; do run QUAD_FRAC=$v OSWALD_HIP_QUAD_FRAC=$v; done
run NO_PRIO OSWALD_HIP_NO_PRIO=1
run TWO_ENDED OSWALD_HIP_TWO_ENDED=1
run ONE_STREAM OSWALD_HIP_ONE_STREAM=1
for v in 5 20; do run COL_COST=$v OSWALD_HIP_COL_COST=$v; done
ONLY_RANK=0 python tests/shard_balance_gpu.py 8 2>&1 | grep -o "shard ms \[[0-9.]*\]"
