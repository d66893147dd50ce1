# Compile-time ablations of the ring GEMM (profiles/round5_gemm_ring_ablation.txt).  The variants are built here with
#   for v in "a7:-DRING_ABLATE=7" "a7nb:-DRING_ABLATE=7 -DRING_NOBAR" "a7nw:-DRING_ABLATE=7 -DRING_NOWAIT" \
#            "a7nbnw:-DRING_ABLATE=7 -DRING_NOBAR -DRING_NOWAIT" "nb:-DRING_NOBAR" "a1:-DRING_ABLATE=1" "a2:-DRING_ABLATE=2" \
#            "a6:-DRING_ABLATE=6" "a5:-DRING_ABLATE=5"; do python tools/variant.py ${v%%:*} gemm_ring.hip "${v#*:} -Wno-c++20-extensions"; done
# (RING_ABLATE bits: 1 no MFMA, 2 no operand requests, 4 no fragment reads; such a library reports ifseg_experimental_build() != 0
# and bench.py refuses it) and preloaded in front of the laboratory binary on the GPU box:
for v in none a7 a7nb a7nw a7nbnw nb a1 a2 a6 a5; do
  echo "== variant $v"
  if [ $v = none ]; then P=""; else P="$GRAFT_REPO_ROOT/ifseg_amd/lib/variants/$v.so"; fi
  LD_PRELOAD="/opt/rocm/lib/libamdhip64.so $P" timeout 60 tools/bin/gemm_lab one 10 "1,6,3,5" 0 8480 768 3072 | grep ring
  LD_PRELOAD="/opt/rocm/lib/libamdhip64.so $P" timeout 60 tools/bin/gemm_lab one 10 "1,6,3,5" 0 8480 3072 768 | grep ring
done
