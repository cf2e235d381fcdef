#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tee $O/parity_5.log | tail -3
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for v in 0 1; do OSWALD_HIP_PAIR_TAILS=$v python bench.py --workload hi --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"hi tails=$v:\", d[\"value\"], d[\"ms_per_step\"], d[\"rerun_ms_per_step\"], d[\"rerun_items_int32\"])"; done
for v in 0 1; do OSWALD_HIP_PAIR_TAILS=$v python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"c4 tails=$v:\", d[\"value\"], d[\"ms_per_step\"], d[\"inclusive\"][\"value\"])"; done
python bench.py --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"c2:\", d[\"value\"], d[\"ms_per_step\"], d[\"inclusive\"][\"value\"])"
python bench.py --workload c5 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"c5:\", d[\"value\"], d[\"ms_per_step\"], d[\"inclusive\"][\"value\"])"
