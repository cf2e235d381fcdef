#!/bin/bash
# tools/profile_gpu.sh -- rocprofv3 evidence for bench.py's dominant kernel, run
# on the GPU box through gpurun:  gpurun -- 'bash tools/profile_gpu.sh c2 100000'
# Pass 1: kernel trace + stats (durations).  Passes 2-4: PMC counters, each in
# its own run with nothing but --pmc (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; counters are never mixed with trace domains).
# Raw output goes to gpurun_out/prof_*; tools/summarize_prof.py condenses it.
set -e -o pipefail
WL=${1:-c2}
NSEQ=${2:-100000}
REPO=$(pwd)
# one directory per (workload, database size, extra label), emptied first: two runs can never share raw files
# (round 3 keyed it by workload only; the C2 summaries at 100 000 and 1 000 000 sequences then mixed their traces)
OUT=$REPO/gpurun_out/prof_${WL}_${NSEQ}${PROF_LABEL:+_$PROF_LABEL}
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --workload $WL --nseq $NSEQ --cpu-seconds 0 ${BENCH_EXTRA:-}"  # e.g. BENCH_EXTRA="--cell-bits 11"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 5 --warmup 1 > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH --steps 2 --warmup 0 > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH --steps 2 --warmup 0 > "$OUT/write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq" -- $BENCH --steps 2 --warmup 0 > "$OUT/sq.log" 2>&1
echo "sq pass done"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$OUT/sq2" -- $BENCH --steps 2 --warmup 0 > "$OUT/sq2.log" 2>&1 || echo "sq2 pass failed (counters not available?)"
echo "sq2 pass done"
# effective clock of the dispatches: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md, DVFS give-back)
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/clk" -- $BENCH --steps 2 --warmup 0 > "$OUT/clk.log" 2>&1 || echo "clk pass failed"
echo "clk pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- $REPO/tools/ubench calib > "$OUT/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -- $REPO/tools/ubench calib > "$OUT/calib_write.log" 2>&1
echo "calibration passes done"
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" "$WL" "$NSEQ" > "$OUT/summary.txt" 2>&1 || true
cat "$OUT/summary.txt"
