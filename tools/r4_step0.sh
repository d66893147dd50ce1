#!/bin/bash
# Round 4, step 0: the bound of removing every per-batch bias term from the attention kernels (wrong results, timing only)
# next to the unmodified step; the attention kernels alone on the encoder shape for both libraries.
out=gpurun_out/r4_step0.txt; : > $out
b() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %7.2f img/s  %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" >> $out 2>&1; }
b baseline
IFSEG_LIB=$PWD/ifseg_amd/lib/variants/nobias_bwd.so b nobias_bwd
IFSEG_LIB=$PWD/ifseg_amd/lib/variants/nobias_all.so b nobias_fwd_bwd
b baseline_again
for k in enc dec cross; do python tools/attn_bench.py $k >> $out 2>&1; done
echo "--- nobias_bwd library" >> $out
for k in enc dec cross; do IFSEG_LIB=$PWD/ifseg_amd/lib/variants/nobias_all.so python tools/attn_bench.py $k >> $out 2>&1; done
cat $out
