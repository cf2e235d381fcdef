#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -m gpu -x -q > $O/t3.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/t3.txt
OSWALD_HIP_DEBUG_SLOW=1 timeout -k 10 300 python tools/inclusive_probe.py 1000000 375 > $O/inclusive_probe_q1.txt 2>&1; echo "probe q1 rc=$?"; grep "inclusive pass\|resident pass" $O/inclusive_probe_q1.txt
OSWALD_HIP_DEBUG_SLOW=1 timeout -k 10 300 python tools/inclusive_probe.py 1000000 > $O/inclusive_probe_c4.txt 2>&1; echo "probe c4 rc=$?"; grep "inclusive pass\|resident pass" $O/inclusive_probe_c4.txt
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e 375 > $O/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"; grep "timed region\|Search speed" $O/cli_q1_1m.txt
for i in 1 2 3; do OSWALD_HIP_DEBUG_SLOW=1 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db 2> $O/cli_q1_1m_dbg$i.err | grep "Search speed"; done
for i in 1 2 3; do oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db | grep "Search speed"; done
for i in 1 2; do oswald_amd/oswald -O search -m 1 -c 16 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db | grep "Search speed"; done
