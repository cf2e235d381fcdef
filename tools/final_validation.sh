#!/bin/bash
# tools/final_validation.sh ROUND [prof|prof2|bench|cli] -- the round's evidence on the GPU box, in parts of at most 20 minutes (one gpurun
# call each).  Everything lands under gpurun_out/final_<part>/ (scratch) with the FULL log of every step in a file of its own (round 6: a run
# whose output had gone through `tail` into a name the next call overwrote left an abort unexplained); tools/collect_final.py ROUND copies
# what the documents cite into profiles/.  The profile parts come first: they write the traffic files of the sources being validated
# (copy gpurun_out/prof_*/traffic_*.json into profiles/ in the build container before `bench`, so that its lines carry roofline.traffic).
R=${1:-06}
PART=${2:-bench}
OUT=gpurun_out/final_$PART
rm -rf $OUT; mkdir -p $OUT
prof() { local name=$(echo "$@" | tr ' =' '__'); bash tools/profile_gpu.sh "$@" > $OUT/prof_$name.log 2>&1; echo "profile $* rc=$?"; }
b() { name=$1; shift; python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;l=[x for x in open('$OUT/bench_$name.json').read().split(chr(10)) if x.startswith('{')];d=json.loads(l[-1]);print(d['value'],d['ms_per_step'],'kernel',d['roofline']['kernel_ms'],'traffic',d['roofline']['traffic'],'inclusive',d['inclusive']['value'],d.get('top_equals_single_gpu_reference_run'),d.get('top10_equals_oracle'),(d.get('cpu_baseline') or {}).get('gpu_scores_equal_on_sample'))" 2>/dev/null)"; }
case $PART in
prof)
  prof c2 1000000
  OSWALD_HIP_PAIR_TAILS=0 PROF_LABEL=tails0 prof c2 1000000
  prof c2 100000
  ;;
prof2)
  prof c3 100000
  prof c5 100000
  prof q1 100000
  prof q1 1000000
  ;;
bench)
  python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
  b c4_1gpu --steps 20 --warmup 5
  OSWALD_HIP_PAIR_TAILS=0 b c4_1gpu_notails --steps 20 --warmup 5 --cpu-seconds 0
  b c4_1gpu_per_chunk --steps 20 --warmup 5 --cpu-seconds 0 --per-chunk-launches
  b q1_1m_per_chunk --workload q1 --nseq 1000000 --steps 20 --warmup 5 --cpu-seconds 0 --per-chunk-launches
  b c4_1gpu_comm --steps 20 --warmup 5 --comm --cpu-seconds 0
  b c2 --nseq 100000 --steps 20 --warmup 5
  b c3_int8 --workload c3 --steps 10 --warmup 2
  b c3_int16 --workload c3 --cell-bits 16 --steps 20 --warmup 5 --cpu-seconds 0
  b c5 --workload c5 --steps 20 --warmup 5
  b c5_1m --workload c5 --nseq 1000000 --steps 5 --warmup 1 --cpu-seconds 0
  b q1_100k --workload q1 --steps 50 --warmup 10
  b q1_1m --workload q1 --nseq 1000000 --steps 20 --warmup 5 --cpu-seconds 0
  b 10m --nseq 10000000 --steps 3 --warmup 1 --cpu-seconds 0
  b hi --workload hi --steps 5 --warmup 1 --cpu-seconds 10
  b hi8 --workload hi8 --steps 8 --warmup 2 --cpu-seconds 10
  b c2_int32 --nseq 100000 --cell-bits 32 --steps 5 --warmup 1 --cpu-seconds 0
  OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29641 python bench.py --gpus 4 --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; echo "gloo4 rc=$?"
  python tests/shard_balance_gpu.py 2 4 8 > $OUT/shard_balance.txt 2>&1; echo "shard balance rc=$?"; cat $OUT/shard_balance.txt
  ;;
cli)
  make -s -C tools probes > $OUT/make_probes.log 2>&1; echo "probes built rc=$?"
  python tools/cli_e2e.py 1000000 /tmp/osw_e2e_1000000 > $OUT/cli_1m.txt 2>&1; echo "cli e2e rc=$?"
  (echo "== three more runs of the tool (each a process of its own)"; for k in 1 2 3; do OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_1000000/q.fasta -d /tmp/osw_e2e_1000000/db 2>&1 >/dev/null | grep "timed region\|device buffers\|host buffers"; done
   echo "== without the warm-up behind the bring-up (OSWALD_HIP_WARM_MS=0)"; for k in 1 2; do OSWALD_HIP_WARM_MS=0 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_1000000/q.fasta -d /tmp/osw_e2e_1000000/db 2>&1 >/dev/null | grep "timed region"; done
   echo "== padded pairs (OSWALD_HIP_PAIR_TAILS=0)"; for k in 1 2; do OSWALD_HIP_PAIR_TAILS=0 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_1000000/q.fasta -d /tmp/osw_e2e_1000000/db 2>&1 >/dev/null | grep "timed region"; done) >> $OUT/cli_1m.txt 2>&1
  grep "timed region\|Search speed" $OUT/cli_1m.txt | head -12
  # ONE 375-residue query -- OSWALD's normal use --: the first pass of a process, four more processes, four passes in one process
  python tools/cli_e2e.py 1000000 /tmp/osw_e2e_q1 375 > $OUT/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"
  (echo "== four more runs of the tool"; for k in 1 2 3 4; do OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region"; done
   echo "== without the warm-up behind the bring-up (OSWALD_HIP_WARM_MS=0)"; for k in 1 2 3 4; do OSWALD_HIP_WARM_MS=0 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region"; done
   echo "== the same search four times in one process (OSWALD_DEBUG_REPEAT=4): the report is the first pass's"; OSWALD_DEBUG_REPEAT=4 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region") >> $OUT/cli_q1_1m.txt 2>&1
  grep "timed region" $OUT/cli_q1_1m.txt | head -14
  (echo "== 1 000 000 sequences, 20 queries (C4 database); $(nproc) hardware threads visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; CS="4 16 64 250" bash tools/hybrid_check.sh 1000000; echo "== 100 000 sequences (C2)"; CS="4 16 64 250" bash tools/hybrid_check.sh 100000) > $OUT/cli_hybrid.txt 2>&1; grep "^-m" $OUT/cli_hybrid.txt
  timeout -k 10 120 ./tools/alloc_probe > $OUT/alloc_probe.txt 2>&1
  timeout -k 10 200 ./tools/oprate4 > $OUT/oprate4.txt 2>&1
  timeout -k 10 100 ./tools/oprate8 > $OUT/oprate8.txt 2>&1
  timeout -k 10 100 ./tools/oprate9 > $OUT/oprate9.txt 2>&1
  timeout -k 10 200 ./tools/oprate_q8 > $OUT/oprate_q8.txt 2>&1
  ;;
esac
