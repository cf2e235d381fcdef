#!/bin/bash
# tools/sweep_granularity.sh -- single-query searches (Q1, C5) at 100 000 and 1 M sequences against the planner's granularity rule
# (OSWALD_HIP_ENTRIES_PER_WG, -DOSW_DIAG build: 0 = off) and, if oswald_amd/liboswald_hip_old.so exists, against that build
OUT=gpurun_out/gran
mkdir -p $OUT
one() { # label, lib env, entries, workload, nseq, steps, warmup
  local f=$OUT/$1_$4_$5.json
  OSWALD_HIP_USE_DIAG_LIB=$2 OSWALD_HIP_ENTRIES_PER_WG=$3 OSWALD_HIP_DEBUG=1 python bench.py --workload $4 --nseq $5 --steps $6 --warmup $7 --cpu-seconds 0 > $f 2> $OUT/$1_$4_$5.err
  echo "$1 $4 $5: $(python -c "import json;d=json.loads(open('$f').read().strip().split(chr(10))[-1]);print(d['value'],'GCUPS',d['ms_per_step'],'ms, kernel',d['roofline']['kernel_gcups'],'entries',d['work_items']//4,'max lg',d['max_log2_geometry'],'spill',d['planned_spill_bytes_per_step'])")  $(grep 'workgroup lg=' $OUT/$1_$4_$5.err | sort -u | sed 's/.*workgroup //' | tr '\n' ';')"
}
for wl in "q1 100000 50 10" "c5 100000 20 5" "q1 1000000 20 5" "c5 1000000 5 1"; do
  set -- $wl
  [ -f oswald_amd/liboswald_hip_old.so ] && one old liboswald_hip_old.so 0 $1 $2 $3 $4
  for e in 0 2 4 8; do one e$e 1 $e $1 $2 $3 $4; done
done
