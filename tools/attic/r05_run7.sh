#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29611 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --nseq 40000 --cpu-seconds 0 --max-chunk 4000000 > $O/gloo2.txt 2>&1; echo "gloo2 rc=$?"; grep -v "^\s*$" $O/gloo2.txt | grep -iv "warn" | head -40
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
