#!/bin/bash
# round 6, call 11: one-launch input checks (no torch glue between the steps) -- tests, A/B against the previous commit's tree
o=gpurun_out/r6_call11; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -k "refused or bad_label or out_of or padded or fixture_forward or deferred or label" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
REPS=4 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-40 $o/ab.txt
