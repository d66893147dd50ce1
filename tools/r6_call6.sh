#!/bin/bash
# round 6, call 6: three cheap experiments on the grouped weight-gradient GEMM (laboratory switches) + the two fixed tests
o=gpurun_out/r6_call6; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -q -x -s -k "resized or gemm_tn or attn_bwd_batch_inner" > $o/pytest.txt 2>&1; tail -4 $o/pytest.txt
grep -n "resized-grid training\|gradients behind" $o/pytest.txt
IFSEG_LAB=1 IFSEG_DW_MFAST=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn" > $o/pytest_mfast.txt 2>&1; tail -2 $o/pytest_mfast.txt
python tools/dwgroup_bench.py > $o/dwgroup.txt 2>&1; IFSEG_LAB=1 IFSEG_DW_MFAST=1 python tools/dwgroup_bench.py > $o/dwgroup_mfast.txt 2>&1; paste $o/dwgroup.txt $o/dwgroup_mfast.txt | grep cap
REPS=3 STEPS=30 bash tools/r6_ab.sh "base:" "mfast:IFSEG_DW_MFAST=1" "even:IFSEG_DW_EVEN=1" "mfast+even:IFSEG_DW_MFAST=1,IFSEG_DW_EVEN=1" "dw_on_main:IFSEG_DW_ON_MAIN=1" > $o/ab.txt 2>&1
cut -c1-50 $o/ab.txt
