#!/bin/bash
# round 6, call 23: 1x1 stride-1 convolutions of the trunk through the plain NT GEMM path (scalar-offset loader) -- tests, A/B, conv_bench
o=gpurun_out/r6_call23; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or stem or gemm" > $o/pytest_k.txt 2>&1; tail -2 $o/pytest_k.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -x -k "fixture_forward or base_config1 or large or resnet or eval or determin" > $o/pytest_model.txt 2>&1; tail -2 $o/pytest_model.txt
REPS=5 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-120 $o/ab.txt
python tools/conv_bench.py > $o/conv_bench.txt 2>&1; tail -14 $o/conv_bench.txt
(cd tools/bin/base && python tools/conv_bench.py) > $o/conv_bench_base.txt 2>&1; tail -14 $o/conv_bench_base.txt
