#!/bin/bash
# round 6, call 1: the -m gpu suite on the tree as it stands (oracle pins, guarded entries), then the default bench line
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/suite_1.log | tail -5
echo "pytest rc=${PIPESTATUS[0]}"
timeout -k 10 300 python bench.py > $O/bench_default_1.json 2> $O/bench_default_1.err; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06/bench_default_1.json').read().split('\n') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['value_inclusive'], d.get('top10_equals_oracle'), d.get('top10_oracle'), d['top_equals_single_gpu_reference_run'])
PY
