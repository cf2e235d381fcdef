#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e 375 > $O/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"; grep "timed region\|Search speed" $O/cli_q1_1m.txt
OSWALD_DEBUG_REPEAT=4 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db 2> $O/cli_q1_1m_repeat.err | grep "Search speed"; grep "timed region" $O/cli_q1_1m_repeat.err
OSWALD_HIP_DEBUG_SLOW=1 OSWALD_DEBUG_REPEAT=3 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db 2> $O/cli_q1_1m_repeat_dbg.err | grep "Search speed"
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e20 > $O/cli_c4_1m.txt 2>&1; echo "cli c4 rc=$?"; grep "timed region\|Search speed" $O/cli_c4_1m.txt
for m in 0 1; do for i in 1 2; do oswald_amd/oswald -O search -m $m -c 16 -q /tmp/osw_e2e20/q.fasta -d /tmp/osw_e2e20/db | grep "Search speed"; done; done
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('default: value', d['value'], d['ms_per_step'], 'inclusive', d['inclusive']['value'], d['inclusive']['ms_per_step'], d['inclusive']['vs_resident'], 'pageable', d['pcie_inclusive_pageable'], 'cpu eq', d['cpu_baseline']['gpu_scores_equal_on_sample'])
"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
