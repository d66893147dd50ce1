#!/bin/bash
for v in "" bi_nodma bi_nocompute; do
  echo "== variant: ${v:-none}"
  if [ -n "$v" ]; then export IFSEG_LIB=$PWD/ifseg_amd/lib/variants/$v.so; fi
  python tools/attn_bi_bench.py enc 2>&1 | grep "bi d"
done
