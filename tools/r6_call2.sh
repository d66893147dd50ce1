#!/bin/bash
# round 6, call 2: LayerNorm-backward pairs -- kernel test, model parity tests, same-box A/B
o=gpurun_out/r6_call2; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln_bwd or layernorm" > $o/pytest_ln.txt 2>&1; tail -3 $o/pytest_ln.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "fixture_forward_backward or base_config1 or deterministic or padded or dropout" > $o/pytest_model.txt 2>&1; tail -3 $o/pytest_model.txt
REPS=2 STEPS=30 bash tools/r6_ab.sh "pairs:" "nopairs:IFSEG_NO_LN_BWD_PAIRS=1" > $o/ab.txt 2>&1
cut -c1-60 $o/ab.txt
python tools/ln_bench.py > $o/ln_bench.txt 2>&1; tail -8 $o/ln_bench.txt
