#!/bin/bash
# round 6, last call on the final sources: the cli part, the shard-balance prediction once more (one rank's shard took 52 instead of 41 ms in
# the bench part's run: both files are kept), a soak and a fuzz campaign
O=gpurun_out/r06; mkdir -p $O
bash tools/final_validation.sh 06 cli
python tests/shard_balance_gpu.py 2 4 8 > $O/shard_balance_2.txt 2>&1; echo "shard balance rc=$?"; grep "deal.*world 8" $O/shard_balance_2.txt | cut -c1-260
( timeout -k 10 400 python tests/soak_gpu.py 300 2>&1 | tee $O/soak_final.log | awk 'NR % 60 == 0 || /done/' ); echo "soak rc=${PIPESTATUS[0]}"
OSWALD_FUZZ_SEED=611 OSWALD_FUZZ_EXAMPLES=600 timeout -k 10 420 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tee $O/fuzz_final.log | tail -2
(echo "== final sources (two pair-kernel variants, a re-run entry per sequence): tests/soak_gpu.py 300"; tail -2 $O/soak_final.log; echo "== fuzz OSWALD_FUZZ_SEED=611 OSWALD_FUZZ_EXAMPLES=600"; tail -2 $O/fuzz_final.log) > $O/soak_fuzz_final.txt
