#!/bin/bash
for v in "" a1; do
  OSWALD_HIP_USE_DIAG_LIB=${v:+liboswald_hip_$v.so} timeout -k 10 300 python bench.py --nseq 100000 --cell-bits 32 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant [$v]:', d['value'], d['ms_per_step'], d.get('top_equals_single_gpu_golden'))"
done
