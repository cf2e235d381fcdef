#!/bin/bash
# round 6, call 4: host staging before / device buffers inside the clock, warm-up at bring-up: suite, CLI at 1 M (20 queries, one query), bench
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/suite_4.log | tail -4
[ ${PIPESTATUS[0]} -eq 0 ] || { echo "suite failed"; exit 1; }
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e_1m > $O/cli_1m_4.txt 2>&1; echo "cli 1m rc=$?"
grep "device buffers\|host buffers\|timed region\|Search speed\|bring-up" $O/cli_1m_4.txt
for k in 1 2 3; do OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_1m/q.fasta -d /tmp/osw_e2e_1m/db 2>&1 >/dev/null | grep "timed region"; done | tee -a $O/cli_1m_4.txt
echo "-- without the warm-up" | tee -a $O/cli_1m_4.txt
for k in 1 2; do OSWALD_HIP_WARM_MS=0 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_1m/q.fasta -d /tmp/osw_e2e_1m/db 2>&1 >/dev/null | grep "timed region"; done | tee -a $O/cli_1m_4.txt
timeout -k 10 300 python tools/cli_e2e.py 1000000 /tmp/osw_e2e_q1 375 > $O/cli_q1_1m_4.txt 2>&1; echo "cli q1 rc=$?"
grep "device buffers\|host buffers\|timed region\|Search speed" $O/cli_q1_1m_4.txt
for k in 1 2 3 4; do OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region"; done | tee -a $O/cli_q1_1m_4.txt
echo "-- without the warm-up" | tee -a $O/cli_q1_1m_4.txt
for k in 1 2 3 4; do OSWALD_HIP_WARM_MS=0 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region"; done | tee -a $O/cli_q1_1m_4.txt
echo "-- the same search four times in one process" | tee -a $O/cli_q1_1m_4.txt
OSWALD_DEBUG_REPEAT=4 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region" | tee -a $O/cli_q1_1m_4.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/bench_default_4.json 2> $O/bench_default_4.err; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06/bench_default_4.json').read().split('\n') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], 'incl', d['inclusive']['value'], d['inclusive']['ms_per_step'], d.get('top10_equals_oracle'), d['top_equals_single_gpu_reference_run'])
PY
