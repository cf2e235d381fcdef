#!/bin/bash
# tools/final_validation.sh [ROUND] -- the round's evidence in one GPU call: the -m gpu suite, smoke, the bench lines the
# documents quote, the multi-rank rehearsal, the shard-balance prediction, the escalation workloads, rocprofv3 summaries,
# the microbenchmarks.  Everything lands under gpurun_out/final/ (scratch); tools/collect_final.py ROUND copies what is
# cited into profiles/.  The profile passes come FIRST: they write the traffic files of the sources being validated, so
# that the bench lines behind them carry roofline.traffic.
# tools/final_validation.sh ROUND prof | bench | cli: the three parts as three GPU calls (a call is limited to 20 minutes); after `prof`,
# copy gpurun_out/prof_*/traffic_*.json into profiles/ in the build container so that the snapshot of `bench` holds them.
# The tool binaries and the diag library come from `make -C tools` (build container).
R=${1:-05}
PART=${2:-all}
OUT=gpurun_out/final
if [ "$PART" != bench ] && [ "$PART" != cli ]; then rm -rf $OUT; fi
mkdir -p $OUT
if [ "$PART" != bench ] && [ "$PART" != cli ]; then
for spec in "c2 1000000" "c2 100000" "c3 100000" "c5 100000" "q1 100000" "q1 1000000"; do
  bash tools/profile_gpu.sh $spec > $OUT/prof_$(echo $spec | tr ' ' '_').log 2>&1; echo "profile $spec rc=$?"
  cp gpurun_out/prof_$(echo $spec | tr ' ' '_')/traffic_*.json profiles/ 2>/dev/null   # (in this box's copy of the tree: bench.py below reads them)
done
fi
if [ "$PART" = prof ]; then exit 0; fi
if [ "$PART" != cli ]; then
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
b() { name=$1; shift; python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$? $(python -c "import json;l=[x for x in open('$OUT/bench_$name.json').read().split(chr(10)) if x.startswith('{')];d=json.loads(l[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['traffic'],d.get('top_equals_single_gpu_golden'))" 2>/dev/null)"; }
b c4_1gpu --steps 20 --warmup 5
b c4_1gpu_comm --steps 20 --warmup 5 --comm --cpu-seconds 0
OSWALD_HIP_PAIR_TAILS=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_c4_1gpu_notails.json 2> $OUT/bench_c4_1gpu_notails.err; echo "bench c4 without tails rc=$?"
b c2 --nseq 100000 --steps 20 --warmup 5
b c3_int8 --workload c3 --steps 10 --warmup 2
b c3_int16 --workload c3 --cell-bits 16 --steps 20 --warmup 5 --cpu-seconds 0
b c5 --workload c5 --steps 20 --warmup 5
b c5_1m --workload c5 --nseq 1000000 --steps 5 --warmup 1 --cpu-seconds 0
b q1_100k --workload q1 --steps 50 --warmup 10
b q1_1m --workload q1 --nseq 1000000 --steps 20 --warmup 5 --cpu-seconds 0
b q1_10m --workload q1 --nseq 10000000 --steps 5 --warmup 1 --cpu-seconds 0
b 10m --nseq 10000000 --steps 3 --warmup 1 --cpu-seconds 0
b hi --workload hi --steps 5 --warmup 1 --cpu-seconds 10
b hi8 --workload hi8 --steps 8 --warmup 2 --cpu-seconds 10
b c2_int32 --nseq 100000 --cell-bits 32 --steps 5 --warmup 1 --cpu-seconds 0
b c4_int32 --nseq 1000000 --cell-bits 32 --steps 3 --warmup 1 --cpu-seconds 0
OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29641 python bench.py --gpus 4 --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; echo "gloo4 rc=$?"
OSWALD_BENCH_BACKEND=gloo MASTER_PORT=29642 python bench.py --gpus 4 --shard-rule reference --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_gloo4_reference_rule.json 2> $OUT/bench_gloo4_reference_rule.err; echo "gloo4 reference rule rc=$?"
python tests/shard_balance_gpu.py 2 4 8 > $OUT/shard_balance.txt 2>&1; echo "shard balance rc=$?"; cat $OUT/shard_balance.txt
tools/q1_tail.sh > $OUT/q1_tail.txt 2>&1; echo "q1 tail rc=$?"
timeout -k 10 200 ./tools/oprate_q8 > $OUT/oprate_q8.txt 2>&1
timeout -k 10 200 ./tools/oprate4 > $OUT/oprate4.txt 2>&1
timeout -k 10 100 ./tools/oprate5 > $OUT/oprate5.txt 2>&1
timeout -k 10 100 ./tools/oprate6 > $OUT/oprate6.txt 2>&1
timeout -k 10 100 ./tools/oprate7 > $OUT/oprate7.txt 2>&1
timeout -k 10 100 ./tools/oprate8 > $OUT/oprate8.txt 2>&1
timeout -k 10 100 ./tools/oprate9 > $OUT/oprate9.txt 2>&1
fi
if [ "$PART" = bench ]; then exit 0; fi
python tools/cli_e2e.py 1000000 /tmp/osw_e2e_1000000 > $OUT/cli_1m.txt 2>&1; echo "cli e2e rc=$?"; tail -5 $OUT/cli_1m.txt
# (round 5) the same tool on ONE 375-residue query -- OSWALD's normal use --, the first pass of a process and three later ones
python tools/cli_e2e.py 1000000 /tmp/osw_e2e_q1 375 > $OUT/cli_q1_1m.txt 2>&1; echo "cli q1 rc=$?"
(echo; echo "== the same search four times in one process (OSWALD_DEBUG_REPEAT=4): the report is the first pass's"; OSWALD_DEBUG_REPEAT=4 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region";
 echo "== pageable residues (OSWALD_NO_PIN=1: round 4's way)"; OSWALD_NO_PIN=1 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "timed region";
 echo "== the device's time line of a pass (OSWALD_HIP_DEBUG_SLOW=1)"; OSWALD_HIP_DEBUG_SLOW=1 OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 0 -q /tmp/osw_e2e_q1/q.fasta -d /tmp/osw_e2e_q1/db 2>&1 >/dev/null | grep "oswald_hip\|timed region") >> $OUT/cli_q1_1m.txt 2>&1
grep "timed region" $OUT/cli_q1_1m.txt | head -8
(echo "== 1 000 000 sequences, 20 queries (C4 database); $(nproc) hardware threads visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; CS="4 16 64 128 250" bash tools/hybrid_check.sh 1000000; echo "== 100 000 sequences (C2)"; CS="4 16 64 128 250" bash tools/hybrid_check.sh 100000) > $OUT/cli_hybrid.txt 2>&1; grep "^-m" $OUT/cli_hybrid.txt
OSWALD_HIP_DEBUG_SLOW=1 python tools/inclusive_probe.py 1000000 375 > $OUT/inclusive_probe_q1.txt 2>&1
OSWALD_HIP_DEBUG_SLOW=1 python tools/inclusive_probe.py 1000000 > $OUT/inclusive_probe_c4.txt 2>&1
timeout -k 10 200 tools/pin_probe 384 /tmp > $OUT/pin_probe.txt 2>&1
tools/startup_probe.sh > $OUT/startup.txt 2>&1
