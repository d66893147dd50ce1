#!/bin/bash
for v in "" bi_NO_SMFMA bi_NO_EXP bi_NO_GMFMA bi_NO_MFMA bi_NO_MFMA_NODMA bi_nodma bi_nocompute; do
  echo "== variant: ${v:-none}"
  if [ -n "$v" ]; then export IFSEG_LIB=$PWD/ifseg_amd/lib/variants/$v.so; fi
  python tools/attn_bi_bench.py enc 2>&1 | grep "bi dkv  "
done
