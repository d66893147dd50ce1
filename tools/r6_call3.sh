#!/bin/bash
# round 6, call 3: the grouped dW through small-LDS 256x128 / 128x256 ring configurations (VERDICT r5 item 1 b)
o=gpurun_out/r6_call3; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ring_grouped" > $o/pytest_ring.txt 2>&1; tail -3 $o/pytest_ring.txt
python tools/dwgroup_bench.py > $o/dwgroup_bench.txt 2>&1; cat $o/dwgroup_bench.txt
REPS=2 STEPS=30 bash tools/r6_ab.sh "base:IFSEG_NO_LN_BWD_PAIRS=1" "ring8:IFSEG_NO_LN_BWD_PAIRS=1,IFSEG_GEMM_RING_GROUP=8" "ring9:IFSEG_NO_LN_BWD_PAIRS=1,IFSEG_GEMM_RING_GROUP=9" "ring10:IFSEG_NO_LN_BWD_PAIRS=1,IFSEG_GEMM_RING_GROUP=10" "ring3:IFSEG_NO_LN_BWD_PAIRS=1,IFSEG_GEMM_RING_GROUP=3" > $o/ab.txt 2>&1
cut -c1-60 $o/ab.txt
