// Launchers of the persistent ring GEMM (gemm_ring.hip), called from the C-ABI entry points in gemm.hip.
#pragma once
// gemm_args: a GemmArgs (gemm_common.h); amode: A_KC / A_KS / A_CONV; cfg: tile configuration id (gemm_ring.hip)
int gemm_ring_launch(const void* gemm_args, int amode, int b_ks, int epi_gln, int cfg, void* stream);
// group_args: a GroupArgs whose start[] / total are recomputed for the configuration's tile shape
int gemm_ring_group_launch(const void* group_args, int cfg, int max_workgroups, void* stream);
