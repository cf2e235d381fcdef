#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
rocm-smi --showpower --showclocks --json 2>&1 | head -c 1500; echo
make -s -C tools probes > $O/make_probes.log 2>&1; echo "probes rc=$?"
bash tools/power_sample.sh q1_1m -- python bench.py --workload q1 --nseq 1000000 --steps 300 --warmup 5 --cpu-seconds 0
bash tools/power_sample.sh c4 -- python bench.py --steps 15 --warmup 3 --cpu-seconds 0
bash tools/power_sample.sh c5_1m -- python bench.py --workload c5 --nseq 1000000 --steps 30 --warmup 3 --cpu-seconds 0
bash tools/power_sample.sh oprate8 -- ./tools/oprate8 20000
bash tools/final_validation.sh 06 prof2
