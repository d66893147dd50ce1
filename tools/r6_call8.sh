#!/bin/bash
# round 6, call 8: tall reduction of the loss statistics + no per-step fill of the dlogits padding (+ early decoder dense) vs HEAD~
o=gpurun_out/r6_call8; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "colsum or fused_seg_loss or fixture_forward_backward or base_config1 or deterministic or label_smoothing" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
REPS=3 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
