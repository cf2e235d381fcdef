#!/bin/bash
b() { name=$1; shift; timeout -k 10 400 python bench.py "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); i=d['inclusive']; print('$name: value', d['value'], d['ms_per_step'], 'inclusive', i['value'], i['ms_per_step'], i['ms_best'], i['vs_resident'])"; }
b c5 --workload c5 --steps 20 --warmup 5 --cpu-seconds 0
b c5_1m --workload c5 --nseq 1000000 --steps 5 --warmup 1 --cpu-seconds 0
b q1_100k --workload q1 --steps 50 --warmup 10 --cpu-seconds 0
b c2 --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0
b c4 --steps 10 --warmup 2 --cpu-seconds 0
b hi --workload hi --steps 5 --warmup 1 --cpu-seconds 0
OSWALD_HIP_SPLIT_BYTES=0 b c2_nosplit --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0
OSWALD_HIP_SPLIT_BYTES=0 b c4_nosplit --steps 10 --warmup 2 --cpu-seconds 0
