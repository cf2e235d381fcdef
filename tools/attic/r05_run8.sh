#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $O/t4.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/t4.txt
(echo "== 1 000 000 sequences, 20 queries (C4 database); $(nproc) hardware threads visible, cpu.max $(cat /sys/fs/cgroup/cpu.max)"; CS="4 16 64 128 250" bash tools/hybrid_check.sh 1000000; echo "== 100 000 sequences (C2)"; CS="4 16 64 128 250" bash tools/hybrid_check.sh 100000) > $O/cli_hybrid.txt 2>&1; grep "^-m\|^==\|hybrid:" $O/cli_hybrid.txt
