run() { python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], 'incl', d['pcie_inclusive']['ms'], 'pageable', round(d['pcie_inclusive_pageable'],1))"; }
for cfg in "" "OSWALD_HIP_PLAN_WAITS=1" "OSWALD_HIP_NO_STREAM_CLASSES=1" "OSWALD_HIP_PLAN_WAITS=1 OSWALD_HIP_NO_STREAM_CLASSES=1"; do
  echo "== [$cfg]"
  for i in 1 2; do env $cfg bash -c "$(declare -f run); run --nseq 100000 --steps 10 --warmup 3"; done
  env $cfg bash -c "$(declare -f run); run --steps 4 --warmup 2"
done
