#!/bin/bash
# round 6, call 29: s_setprio around the MFMA clusters of the attention FORWARD only (four waves per SIMD there)
o=gpurun_out/r6_call29; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn_fwd or attn_bwd_batch_inner" > $o/pytest_attn.txt 2>&1; tail -2 $o/pytest_attn.txt
REPS=8 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
