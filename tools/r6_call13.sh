#!/bin/bash
# round 6, call 13: the one-pass sum_b dS kernel at 168 VGPRs / three waves per SIMD -- A/B against the previous commit; kernel tests
o=gpurun_out/r6_call13; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn_bwd_batch_inner" > $o/pytest_attn.txt 2>&1; tail -3 $o/pytest_attn.txt
REPS=4 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
for k in enc dec; do python tools/attn_bi_bench.py $k 2>&1 | grep -i "dbias\|operands\|tables" ; done > $o/attn_bi_bench.txt; cat $o/attn_bi_bench.txt
