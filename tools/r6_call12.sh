#!/bin/bash
# round 6, call 12: one pass over sum_b dS for the d pos_q operand AND the 2-D table partials (second half) -- kernel
# tests, model parity, A/B against the previous commit; round filling of the N = 768 products (item 2b) alone and in the step
o=gpurun_out/r6_call12; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn_bwd_batch_inner" > $o/pytest_attn.txt 2>&1; tail -3 $o/pytest_attn.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -k "fixture_forward or base_config1 or determin or resize or image_free" > $o/pytest_model.txt 2>&1; tail -3 $o/pytest_model.txt
REPS=3 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
for k in enc dec; do python tools/attn_bi_bench.py $k 2>&1 | grep -i "dbias\|operands\|tables" ; done > $o/attn_bi_bench.txt; cat $o/attn_bi_bench.txt
python tools/r6_roundfill.py > $o/roundfill.txt 2>&1; cat $o/roundfill.txt
for rep in 1 2 3; do
  for t in default narrow; do
    e=""; [ $t = narrow ] && e="IFSEG_GEMM_NARROW_MAX=512"
    out=$(env IFSEG_LAB=1 $e python bench.py --lab --steps 30 --warmup 6 --no-cpu-baseline --steady-steps 0 2>/dev/null | tail -1)
    echo "$t $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done > $o/narrow_in_step.txt; cat $o/narrow_in_step.txt
