#!/bin/bash
# round 6, call 14: the FFN row dots inside the two-output LayerNorm backward (one launch and two main-queue boundaries less per FFN block)
o=gpurun_out/r6_call14; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "ln_bwd_drop or ffn_ln or layernorm" > $o/pytest_k.txt 2>&1; grep -i "fused row dots" $o/pytest_k.txt; tail -2 $o/pytest_k.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -k "fixture_forward or base_config1 or determin or padded or resized or large" > $o/pytest_model.txt 2>&1; tail -3 $o/pytest_model.txt
REPS=4 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
