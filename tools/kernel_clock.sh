#!/bin/bash
# tools/kernel_clock.sh -- the core clock INSIDE the single-query kernel (round 6): the -DOSW_DIAG build of the library
# (`make -C oswald_amd/csrc diag`) sums, over the workgroups of a launch, the cycle counter and the 100-MHz counter between a workgroup's
# first and last instruction; with it the finish times of the launch's workgroups (per-decile tables).  Beside it: what rocprofv3's
# GRBM_GUI_ACTIVE gives for the same kernel ("effective clock" in profiles/r06_*_summary.json: busy cycles over wall time, i.e. a figure
# that also falls when part of the chip has run out of work) and what the card draws (tools/power_sample.sh).
for spec in "q1 1000000" "q1 100000" "c5 1000000" "c5 100000"; do set -- $spec
  echo "== $1 $2 (diag library; its GCUPS are not the product's)"
  OSWALD_HIP_USE_DIAG_LIB=1 OSWALD_HIP_DEBUG_TIMES=1 python bench.py --workload $1 --nseq $2 --steps 2 --warmup 1 --cpu-seconds 0 2>&1 >/dev/null | grep "oswald_hip" | tail -8
done
