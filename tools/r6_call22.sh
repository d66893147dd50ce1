#!/bin/bash
# round 6, call 21: LayerNorm backward at 80 VGPRs (column sums and gains in LDS): two waves per SIMD beside the grouped dW GEMM instead of one
o=gpurun_out/r6_call22; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln or layernorm or dropout or colsum or embed or rows" > $o/pytest_k.txt 2>&1; tail -2 $o/pytest_k.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x > $o/pytest_model.txt 2>&1; tail -2 $o/pytest_model.txt
REPS=3 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-120 $o/ab.txt
python tools/ln_bench.py > $o/ln_bench.txt 2>&1; tail -12 $o/ln_bench.txt
(cd tools/bin/base && python tools/ln_bench.py) > $o/ln_bench_base.txt 2>&1; tail -12 $o/ln_bench_base.txt
