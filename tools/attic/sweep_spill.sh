#!/bin/bash
# tools/sweep_spill.sh -- spill traffic against throughput on C2: which items run as workgroup items (4x taller rounds, 1/4 of
# the round boundaries) -- by block length (WG_MINCOLS) and by the number of rounds of the wave plan (WG_MINROUNDS, from
# WG_MINCOLS_LONG columns on).  Prints GCUPS and the planned spill bytes per search (diag library).
export OSWALD_HIP_USE_DIAG_LIB=1
run() { echo -n "$* : "; env "$@" python bench.py --workload c2 --nseq 100000 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'GCUPS', d['roofline']['kernel_ms'], 'ms', round(d['planned_spill_bytes_per_step']/1e9,2), 'GB planned spill')"; }
run X=1
for mc in 1024 512 256 0; do run OSWALD_HIP_WG_MINCOLS=$mc; done
for mr in 16 12 8 6 4; do for mcl in 0 256 512; do run OSWALD_HIP_WG_MINROUNDS=$mr OSWALD_HIP_WG_MINCOLS_LONG=$mcl; done; done
