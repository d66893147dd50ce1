#!/bin/bash
# round 6, call 24: GEMM epilogue operands read per element (the GELU-LN epilogue's u and gains, the row-dot operand) requested together
# ahead of the arithmetic instead of one serial round trip per (j, rg) group -- tests, A/B, stand-alone gln / rowdot times
o=gpurun_out/r6_call24; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or ffn or gln or rowdot or linear or conv or stem" > $o/pytest_k.txt 2>&1; tail -2 $o/pytest_k.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x > $o/pytest_model.txt 2>&1; tail -2 $o/pytest_model.txt
REPS=5 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-120 $o/ab.txt
python tools/ffn_ln_bench.py > $o/ffn_ln_bench.txt 2>&1; tail -8 $o/ffn_ln_bench.txt
(cd tools/bin/base && python tools/ffn_ln_bench.py) > $o/ffn_ln_bench_base.txt 2>&1; tail -8 $o/ffn_ln_bench_base.txt
