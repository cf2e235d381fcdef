#!/bin/bash
# tools/ab_seq_pause.sh -- the single-query row with an s_nop 0 behind its subtracts (builds liboswald_hip_var{P,A,B}.so: both / first / second)
OUT=gpurun_out/pause; mkdir -p $OUT
for rep in 1 2; do
for v in base P A B; do
  for wl in "q1 1000000 20 5" "c5 1000000 5 2"; do set -- $wl
    if [ $v = base ]; then unset OSWALD_HIP_USE_DIAG_LIB; else export OSWALD_HIP_USE_DIAG_LIB=liboswald_hip_var$v.so; fi
    [ -f oswald_amd/liboswald_hip_var$v.so ] || [ $v = base ] || continue
    python bench.py --workload $1 --nseq $2 --steps $3 --warmup $4 --cpu-seconds 0 > $OUT/${v}_$1.json 2> $OUT/${v}_$1.err
    echo "$v $1: $(python -c "import json;d=json.loads(open('$OUT/${v}_$1.json').read().strip().split(chr(10))[-1]);print(d['value'],d['roofline']['kernel_gcups'])")"
  done
done
done
