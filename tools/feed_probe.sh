# profiles/round5_gemm_small_m_and_ksweep.txt: the same per-workgroup work with fewer workgroups (operand-request-only build
# variants/a5.so, see tools/abl_lab.sh), the full kernels, and time against K for the tile kernel
# is the operand feed (LDS-DMA only build, variants/a5.so) limited per CU or chip-wide?  same per-workgroup work, fewer workgroups
P="/opt/rocm/lib/libamdhip64.so $GRAFT_REPO_ROOT/ifseg_amd/lib/variants/a5.so"
for M in 512 1024 2048 4096 8192 16384; do
  LD_PRELOAD="$P" timeout 60 tools/bin/gemm_lab one 10 "1,6" 0 $M 768 3072 | grep ring
done
echo "== full kernels"
for M in 512 1024 2048 4096 8192 16384; do
  timeout 60 tools/bin/gemm_lab one 10 "0,1,6" 0 $M 768 3072 | grep -v abi
done
echo "== ksweep (tile kernel)"
timeout 120 tools/bin/gemm_lab ksweep 20 0
