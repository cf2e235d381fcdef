#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_bench_contract.py -m gpu -x -q > $O/t1.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/t1.txt
for i in 1 2 3 4 5 6; do timeout -k 10 120 python bench.py --workload q1 --nseq 100000 --steps 30 --warmup 5 --cpu-seconds 0 2>>$O/q1_100k_runs.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('q1 100k run: value', d['value'], 'ms', d['ms_per_step'], 'inclusive', d['inclusive']['value'], d['inclusive']['ms_per_step'], 'best', d['inclusive']['ms_best'], 'pageable', d['pcie_inclusive_pageable'])
" | tee -a $O/q1_100k_runs.txt; done
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e 375 > $O/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"; grep -v "^$" $O/cli_q1_1m.txt | head -60
OSWALD_NO_PIN=1 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db 2> $O/cli_q1_1m_nopin.err | grep "Search speed"; grep "timed region" $O/cli_q1_1m_nopin.err
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('default: value', d['value'], d['ms_per_step'], 'inclusive', d['inclusive'], 'pageable', d['pcie_inclusive_pageable'], 'cpu eq', d['cpu_baseline']['gpu_scores_equal_on_sample'])
"
