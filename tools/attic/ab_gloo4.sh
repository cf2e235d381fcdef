run() { OSWALD_BENCH_BACKEND=gloo MASTER_PORT=$1 python bench.py --gpus 4 --steps 4 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
p=29700
for cfg in "X=1" "OSWALD_HIP_PLAN_WAITS=1" "OSWALD_HIP_NO_STREAM_CLASSES=1"; do
  echo "== [$cfg]"; p=$((p+1)); env $cfg bash -c "$(declare -f run); run $p"
done
