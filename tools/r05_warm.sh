#!/bin/bash
# scratch: first-pass time of the one-query CLI search against OSWALD_HIP_WARM_MS
python tools/cli_e2e.py 1000000 /tmp/osw_e2e_q1 375 > /dev/null 2>&1
for rep in 1 2; do
for w in 0 1 3 8 20; do
  echo "== warm $w ms"
  OSWALD_HIP_WARM_MS=$w OSWALD_DEBUG_PHASES=1 OSWALD_HIP_DEBUG_SLOW=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region\|on the device"
done
done
