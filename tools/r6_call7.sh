#!/bin/bash
# round 6, call 7: the decoder's position operands / dense biases ahead of the encoder layers -- tests, same-box A/B
o=gpurun_out/r6_call7; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests/test_model_gpu.py -q -x -k "fixture_forward_backward or base_config1 or deterministic or padded or resized or graph or image_free or deferred" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
REPS=3 STEPS=30 bash tools/r6_ab.sh "late:IFSEG_NO_EARLY_DEC_DENSE=1" "early:IFSEG_LAB=1" > $o/ab.txt 2>&1
cut -c1-50 $o/ab.txt
