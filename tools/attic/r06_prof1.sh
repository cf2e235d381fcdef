#!/bin/bash
# round 6: rocprofv3 evidence (kernel trace + PMC passes) of the headline workload in both modes, and C2
O=gpurun_out/r06; mkdir -p $O
bash tools/profile_gpu.sh c2 1000000 > $O/prof_c2_1000000.log 2>&1; echo "profile c2 1M rc=$?"
OSWALD_HIP_PAIR_TAILS=0 PROF_LABEL=tails0 bash tools/profile_gpu.sh c2 1000000 > $O/prof_c2_1000000_tails0.log 2>&1; echo "profile c2 1M tails0 rc=$?"
bash tools/profile_gpu.sh c2 100000 > $O/prof_c2_100000.log 2>&1; echo "profile c2 100k rc=$?"
ls gpurun_out/prof_c2_1000000/ gpurun_out/prof_c2_1000000_tails0/ | head -40
