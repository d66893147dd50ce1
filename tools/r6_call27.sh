#!/bin/bash
# round 6, call 27: Adam at 512 blocks in the step (A/B against the previous commit's 2048)
o=gpurun_out/r6_call27; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "adam or optim or trainer or updates" > $o/pytest_k.txt 2>&1; tail -2 $o/pytest_k.txt
REPS=6 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-44 $o/ab.txt
