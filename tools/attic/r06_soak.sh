#!/bin/bash
# round 6: soak + fuzz on the final sources (progress lines keep the call alive)
O=gpurun_out/r06; mkdir -p $O
( timeout -k 10 500 python tests/soak_gpu.py 400 2>&1 | tee $O/soak.log | grep -v "^$" | awk 'NR % 40 == 0 || /mismatch|bad|done|total/' ) ; echo "soak rc=${PIPESTATUS[0]}"
for seed in 601 602; do
  OSWALD_FUZZ_SEED=$seed OSWALD_FUZZ_EXAMPLES=700 timeout -k 10 420 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tee $O/fuzz_$seed.log | tail -3; echo "fuzz $seed rc=${PIPESTATUS[0]}"
done
(echo "== soak: tests/soak_gpu.py 400"; tail -4 $O/soak.log; for seed in 601 602; do echo "== fuzz OSWALD_FUZZ_SEED=$seed OSWALD_FUZZ_EXAMPLES=700"; tail -2 $O/fuzz_$seed.log; done) > $O/soak_fuzz.txt; cat $O/soak_fuzz.txt
