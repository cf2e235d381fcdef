#!/bin/bash
# tools/sweep_q8.sh -- planner sensitivity of the 8-bit mode on C3 (diag library: the sweep knobs exist only there)
export OSWALD_HIP_USE_DIAG_LIB=1
run() { echo "== $*"; env "$@" python bench.py --workload c3 --steps 4 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['work_items'], d['max_log2_geometry'])"; }
run X=1
run OSWALD_HIP_DEBUG=1
for td in 0.6 2 4; do run OSWALD_HIP_TARGET_DIV=$td; done
for lg in 1 2 3 4; do run OSWALD_HIP_FORCE_LG=$lg; done
for cc in 30 60; do run OSWALD_HIP_COL_COST=$cc; done
