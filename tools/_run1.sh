cd /root/repo
unset IFSEG_LIB
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn" 2>&1 | tail -3
for k in enc dec cross; do timeout 120 python tools/attn_bench.py $k 2>&1 | grep "alone\|attn_fwd"; done
