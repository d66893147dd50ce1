#!/bin/bash
# round 6, call 9: the exposed end of the backward (tail v2) -- tests, same-box A/B
o=gpurun_out/r6_call9; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -k "fixture_forward_backward or base_config1 or deterministic or padded or resized or graph or image_free or two_rank or rccl or c3" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
REPS=4 STEPS=30 bash tools/r6_ab.sh "old_tail:IFSEG_NO_TAIL_V2=1" "tail_v2:IFSEG_LAB=1" > $o/ab.txt 2>&1
cut -c1-50 $o/ab.txt
IFSEG_LAB=1 IFSEG_DRAIN_TIMING=1 python bench.py --lab --steps 40 --warmup 6 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v2 wait', d.get('end_of_backward_wait_ms'), d['ms_per_step'])"
IFSEG_LAB=1 IFSEG_NO_TAIL_V2=1 IFSEG_DRAIN_TIMING=1 python bench.py --lab --steps 40 --warmup 6 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old wait', d.get('end_of_backward_wait_ms'), d['ms_per_step'])"
