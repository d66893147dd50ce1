for v in none a7 a7nb a7nw a7nbnw nb a1 a2 a6 a5; do
  echo "== variant $v"
  if [ $v = none ]; then P=""; else P="$GRAFT_REPO_ROOT/ifseg_amd/lib/variants/$v.so"; fi
  LD_PRELOAD="/opt/rocm/lib/libamdhip64.so $P" timeout 60 tools/bin/gemm_lab one 10 "1,6,3,5" 0 8480 768 3072 | grep ring
  LD_PRELOAD="/opt/rocm/lib/libamdhip64.so $P" timeout 60 tools/bin/gemm_lab one 10 "1,6,3,5" 0 8480 3072 768 | grep ring
done
