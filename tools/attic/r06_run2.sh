#!/bin/bash
# round 6, call 2: tails inside the pair launch (no planes): the tails tests first, then the whole suite, then bench lines with / without
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "failures or converted" 2>&1 | tee $O/tails_2.log | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || { echo "tails tests failed"; exit 1; }
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/suite_2.log | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || { echo "suite failed"; exit 1; }
for v in 1 0 2; do
  OSWALD_HIP_PAIR_TAILS=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > $O/bench_c4_tails$v.json 2> $O/bench_c4_tails$v.err; echo "bench tails=$v rc=$?"
done
OSWALD_HIP_PAIR_TAILS=1 timeout -k 10 200 python bench.py --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c2_tails1.json 2> $O/bench_c2_tails1.err
OSWALD_HIP_PAIR_TAILS=0 timeout -k 10 200 python bench.py --nseq 100000 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c2_tails0.json 2> $O/bench_c2_tails0.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/bench_c[24]_tails*.json')):
    l=[x for x in open(f).read().split('\n') if x.startswith('{')]
    if not l: print(f,'no line'); continue
    d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'incl', d['inclusive']['value'], d['inclusive']['ms_per_step'], d['top_equals_single_gpu_reference_run'])
PY
