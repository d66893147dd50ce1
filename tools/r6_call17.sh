#!/bin/bash
# round 6, call 17: instruction / stall counters of the batch-inner attention kernels before (commit 86e6eac, tools/bin/base) and after the
# scalar-offset buffer DMA staging (separate --pmc passes, tools/attn_bi_pmc.sh on the encoder shape)
o=gpurun_out/r6_call17; rm -rf $o; mkdir -p $o
R=$GRAFT_REPO_ROOT
bash tools/attn_bi_pmc.sh enc > $o/pmc_new.txt 2>&1
( cd tools/bin/base && GRAFT_REPO_ROOT=$R/tools/bin/base bash tools/attn_bi_pmc.sh enc ) > $o/pmc_base.txt 2>&1
for k in attn_bi_fwd_kernel attn_bi_dq_kernel attn_bi_dkv_kernel; do
  echo "== $k (base | new)"
  paste <(grep -A19 "^$k" $o/pmc_base.txt | tail -19 | cut -c1-52) <(grep -A19 "^$k" $o/pmc_new.txt | tail -19 | cut -c36-52)
done | tee $o/compare.txt
