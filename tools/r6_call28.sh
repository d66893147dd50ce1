#!/bin/bash
# round 6, call 28: s_setprio(1) around the MFMA clusters of the three batch-inner attention kernels
o=gpurun_out/r6_call28; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "attn" > $o/pytest_attn.txt 2>&1; grep "dbias" $o/pytest_attn.txt | cut -c1-220 | head -24; tail -2 $o/pytest_attn.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -s > $o/pytest_model.txt 2>&1; tail -3 $o/pytest_model.txt
grep -i "rel-L2\|worst" $o/pytest_model.txt | cut -c1-250 | head -60 > $o/parity_numbers.txt
REPS=5 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
for k in enc dec cross; do python tools/attn_bi_bench.py $k 2>&1 | grep -i "bi dq\|bi dkv\|bi fwd\|fwd" ; done > $o/attn_bi_bench.txt; cat $o/attn_bi_bench.txt
(cd tools/bin/base; for k in enc dec cross; do python tools/attn_bi_bench.py $k 2>&1 | grep -i "bi dq\|bi dkv\|bi fwd\|fwd" ; done) > $o/attn_bi_bench_base.txt; cat $o/attn_bi_bench_base.txt
