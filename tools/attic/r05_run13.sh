#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
OSWALD_HIP_DEBUG_SLOW=1 timeout -k 10 300 python tools/inclusive_probe.py 1000000 375 > $O/inclusive_probe_q1.txt 2>&1; grep "inclusive pass\|resident pass" $O/inclusive_probe_q1.txt | tail -6; grep "search [0-9]:" $O/inclusive_probe_q1.txt | tail -8
timeout -k 10 600 python bench.py --workload q1 --nseq 10000000 --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('q1 10m: value', d['value'], d['ms_per_step'], 'inclusive', d['inclusive']['value'], d['inclusive']['ms_per_step'], d['inclusive']['vs_resident'])"
