#!/bin/bash
# round 6: the resident steps as one launch over the rank's chunks (oswald_hip_search_resident) against one launch per chunk
run() { python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'x', d['roofline']['launches_per_step'], 'incl', d['inclusive']['value'], d['top_equals_single_gpu_reference_run'], d['config']['resident_search'][:24])"; }
for extra in "" "--per-chunk-launches"; do
  echo "== $extra"
  echo "  c4";     run --steps 10 --warmup 3 $extra
  echo "  q1 1m";  run --workload q1 --nseq 1000000 --steps 30 --warmup 5 $extra
  echo "  c5 1m";  run --workload c5 --nseq 1000000 --steps 8 --warmup 2 $extra
  echo "  10m";    run --nseq 10000000 --steps 2 --warmup 1 $extra
  echo "  c3 int8 1m"; run --workload c3 --nseq 1000000 --steps 3 --warmup 1 $extra
done
python bench.py --steps 5 --warmup 2 > gpurun_out/r06/bench_resident_default.json 2> gpurun_out/r06/bench_resident_default.err; echo "default rc=$?"
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06/bench_resident_default.json') if l.startswith('{')][-1]); print(d['value'], d['value_inclusive'], d['top10_equals_oracle'], d['roofline']['traffic_note'], d['cpu_baseline']['gpu_scores_equal_on_sample'])"
