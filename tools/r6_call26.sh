#!/bin/bash
# round 6, call 26: Adam at 512 blocks in the step (A/B against the previous commit); grid cap of the (now grid-stride) LayerNorm forward alone
o=gpurun_out/r6_call26; rm -rf $o; mkdir -p $o
( BLOCKS=768 python tools/ln_bench.py | head -3
  for v in 512 768 1024 1536; do echo "cap $v"; BLOCKS=768 IFSEG_LIB=$GRAFT_REPO_ROOT/ifseg_amd/lib/variants/lnf_$v.so python tools/ln_bench.py | head -3; done ) > $o/ln_fwd_grid.txt 2>&1; grep -v amdgpu $o/ln_fwd_grid.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln or layernorm or adam or optim" > $o/pytest_k.txt 2>&1; tail -2 $o/pytest_k.txt
REPS=5 STEPS=30 bash tools/r6_ab2.sh > $o/ab.txt 2>&1; cut -c1-44 $o/ab.txt
