#!/bin/bash
# round 6: a longer soak + two more fuzz campaigns on HEAD
O=gpurun_out/r06; mkdir -p $O
( timeout -k 10 600 python tests/soak_gpu.py 600 2>&1 | tee $O/soak_head.log | awk 'NR % 100 == 0 || /done/' ); echo "soak rc=${PIPESTATUS[0]}"
for seed in 621 622; do
  OSWALD_FUZZ_SEED=$seed OSWALD_FUZZ_EXAMPLES=1000 timeout -k 10 560 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tee $O/fuzz_head_$seed.log | tail -2; echo "fuzz $seed rc=${PIPESTATUS[0]}"
done
(echo "== HEAD: tests/soak_gpu.py 600"; tail -2 $O/soak_head.log; for seed in 621 622; do echo "== fuzz OSWALD_FUZZ_SEED=$seed OSWALD_FUZZ_EXAMPLES=1000"; tail -2 $O/fuzz_head_$seed.log; done) > $O/soak_fuzz_head.txt
