#!/bin/bash
# tools/cli_f2_probe.sh -- `oswald -f 2` with both context devices on the box's one GPU (OSWALD_DEVICE_IDS=0,0) at 1 M sequences:
# host-side phases and any section of a library call that holds the host (OSWALD_HIP_DEBUG_SLOW); -f 1 beside it.
# (A rehearsal of the dealt multi-device flow: index maps, uploads two rounds ahead on two devices; the two devices share the GPU,
# so the searches take turns and the speed is not that of two GPUs.)
T=/tmp/osw_e2e
if [ ! -f $T/db.info ]; then python tools/cli_e2e.py 1000000 $T > /dev/null 2>&1; fi
for f in 1 2; do
  ids=0; [ $f = 2 ] && ids=0,0
  echo "== -f $f"
  OSWALD_DEVICE_IDS=$ids OSWALD_DEBUG_PHASES=1 OSWALD_HIP_DEBUG_SLOW=1 OSWALD_HIP_NO_STREAM_CLASSES=${NOCLASSES:-0} oswald_amd/oswald -O search -m 0 -f $f -q $T/q.fasta -d $T/db 2>&1 >/tmp/osw_f$f.out | grep -E "held|slow|queue|timed region|gather" | cut -c1-150
  grep -E "Search speed" /tmp/osw_f$f.out
done
cmp <(grep -A12 "Query no" /tmp/osw_f1.out | head -60) <(grep -A12 "Query no" /tmp/osw_f2.out | head -60) && echo "reports equal (first queries)"
