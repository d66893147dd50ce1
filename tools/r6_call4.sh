#!/bin/bash
# round 6, call 4: bf16 dense attention bias -- kernel tests, goldens (numbers printed), A/B against the previous commit
o=gpurun_out/r6_call4; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn" > $o/pytest_attn.txt 2>&1; tail -5 $o/pytest_attn.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -s > $o/pytest_model.txt 2>&1; tail -5 $o/pytest_model.txt
grep -i "rel-L2\|rel_l2\|logits\|worst" $o/pytest_model.txt | head -60 > $o/parity_numbers.txt
REPS=3 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
python tools/attn_bi_bench.py > $o/attn_bi_bench.txt 2>&1; tail -12 $o/attn_bi_bench.txt
(cd tools/bin/base && python tools/attn_bi_bench.py) > $o/attn_bi_bench_base.txt 2>&1; tail -12 $o/attn_bi_bench_base.txt
