#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
OSWALD_PROBE_REGISTERED=1 OSWALD_HIP_DEBUG_SLOW=1 timeout -k 10 300 python tools/inclusive_probe.py 1000000 375 > $O/inclusive_probe_q1_reg.txt 2>&1; echo "probe q1 registered rc=$?"; grep "inclusive pass\|held" $O/inclusive_probe_q1_reg.txt | tail -12
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e 375 > $O/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"; grep "timed region\|Search speed" $O/cli_q1_1m.txt
for i in 1 2 3; do OSWALD_HIP_DEBUG_SLOW=1 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db 2> $O/cli_q1_1m_dbg$i.err | grep "Search speed"; grep "held\|started on" $O/cli_q1_1m_dbg$i.err; done
