#!/bin/bash
# tools/q1_tail.sh -- what is left of a single-query search: the default bench lines of Q1 and C5 at
# 100 000 and 1 000 000 sequences, and -- with the -DOSW_DIAG build of the library -- when the workgroups and the CUs of the
# DP launch finished (per-decile tables of OSWALD_HIP_DEBUG_TIMES; the diag build stamps times inside the kernel, its
# GCUPS are not the product's).
for wl in q1 c5; do for n in 100000 1000000; do
  echo "== $wl $n (product library)"
  python bench.py --workload $wl --nseq $n --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('GCUPS', d['value'], 'ms/step', d['ms_per_step'], 'kernel ms', r['kernel_ms'], 'kernel GCUPS', r['kernel_gcups'], 'of the VALU ceiling', r['valu']['frac'], 'inclusive', d['pcie_inclusive']['gcups'], 'work items', d['work_items'])"
  echo "== $wl $n (diag library: finish times of the last step's launches)"
  OSWALD_HIP_USE_DIAG_LIB=1 OSWALD_HIP_DEBUG_TIMES=1 python bench.py --workload $wl --nseq $n --steps 2 --warmup 1 --cpu-seconds 0 2>&1 >/dev/null | grep "oswald_hip" | tail -7
done; done
