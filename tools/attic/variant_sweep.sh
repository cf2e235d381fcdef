#!/bin/bash
# tools/variant_sweep.sh [bench args] -- time every tools/variants/lib_*.so (kernel variants built by hand with -DOSW_NOPMASK=..)
# through bench.py: each is copied over the diagnostics library's path and loaded with OSWALD_HIP_USE_DIAG_LIB=1.
for f in tools/variants/lib_*.so; do
  cp $f oswald_amd/liboswald_hip_diag.so
  OSWALD_HIP_USE_DIAG_LIB=1 python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['top1_scores'][:3])"
done
