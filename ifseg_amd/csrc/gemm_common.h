// Shared definitions of the GEMM kernels (gemm.hip: one tile per workgroup, split-K; conv.hip; tools/gemm_ring/: the persistent
// ring-of-stages laboratory of round 5, not part of the library): argument block, LDS tile images, LDS-DMA pieces.
#pragma once
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr int GBK = 64;
// LDS stages: 2 (64 KiB, two workgroups per CU, loads overlap the MFMAs inside the workgroup) when the
// grid is at most ~2 rounds of that occupancy; 1 (32 KiB, four workgroups per CU overlap each other) for
// the many-tile shapes.  Measured on the SegOFA-Base shapes, tools/gemm_bench.py.
constexpr int TWO_STAGE_MAX_WGS = 1024;
constexpr int BM = 128;   // BN, BK and the LDS stage count are template parameters
enum { A_KC = 0, A_KS = 1, A_CONV = 2 };
constexpr int GEMM_FLAG_MFAST = 0x100;  // (internal, grouped dW) row tiles fastest in the tile order of a wide problem
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond every buffer (< 2 GiB each): the load returns zeros

struct GemmArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  int M, N, K, lda, ldb, ldc;
  const bf16_t* bias; const bf16_t* resid; int ldr;
  float alpha; int alpha_ncols; int flags;
  int cH, cW, cC, cKW, cStride, cPad, cOH, cOW;
  long long sA, sB, sC, sR;
  int splitk, kchunk; long long sCsplit;
  unsigned nrecA, nrecB;   // bytes addressable through the A / B buffer descriptors (per batch)
  // optional row-dot epilogue (attention backward's delta): dot_out[(m / dot_T) * (N/64) + n/64][m % dot_T] =
  // sum over the 64 columns of head n/64 of C[m][n] (as stored in bf16) * dot[m][n]
  const bf16_t* dot; int ldd; float* dot_out; int dot_T;
  int xcd_groups;          // > 0: split-K slices pinned to XCDs (see the kernel), grid = tiles * splitk workgroups in x
  // optional "GELU + LayerNorm backward" epilogue (EPI_GLN, the FFN's ffn_layernorm(gelu(fc1)) on the way back): the GEMM
  // result is dz = d(LN output); the epilogue turns it into du = d(fc1 output) without dz ever reaching HBM:
  //   g = gelu(u), xh = (g - mean_m) rstd_m, du = rstd_m (gamma_n dz - c1_m - xh c2_m) gelu'(u)
  // c1 / c2 = the two row means of the LayerNorm backward, supplied by the caller (ifseg_ffn_ln_rowstats computes them from
  // 768-wide tensors: the row sums over the 3072 columns are linear in dz = dY . W)
  const bf16_t* gln_u; int gln_ldu; const float* gln_gamma; const float* gln_mean; const float* gln_rstd; const float* gln_c;
};

// ---- LDS tile images -------------------------------------------------------
// Tiles are filled by LDS-DMA (buffer_load_dwordx4 ... lds): one wave instruction
// writes 1 KiB, lane l at byte 16 l, so the image is lane-linear and the XOR
// swizzle is applied on the SOURCE address (x_src below) and again on the read
// (x_off); both are the same involution inside a 256-byte line.
//   KC tile: [rows][BK k], k contiguous in global.   BK = 64: kc_off (common.h).
//            BK = 32: 64-byte rows, chunk ^= (row >> 2) & 3.
//   KS tile: [BK k][128 cols], cols contiguous in global: ks_off (common.h).
template <int BK>
__device__ __forceinline__ int kct_off(int r, int c) {
  if constexpr (BK == 64) return kc_off(r, c);
  else return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}
template <int BK>
__device__ __forceinline__ void kct_src(int seg, int l, int& row, int& c) {
  if constexpr (BK == 64) {
    const int line = seg * 4 + (l >> 4), s = (l & 15) ^ (line & 15);
    row = line * 2 + (s >> 3);
    c = s & 7;
  } else {
    row = seg * 16 + (l >> 2);
    c = (l & 3) ^ ((row >> 2) & 3);
  }
}
__device__ __forceinline__ void ks_src(int seg, int l, int& kr, int& col) {
  kr = seg * 4 + (l >> 4);
  const int slot = l & 15;
  col = ((((slot >> 2) ^ (kr & 3)) << 2) | (slot & 3)) * 8;
}
template <int BK>
__device__ __forceinline__ bf16x8 frag_kct(const unsigned char* tile, int rb, int ks, int lane) {
  return lds_read_b128(tile + kct_off<BK>(rb + (lane & 31), ks * 2 + (lane >> 5)));
}

typedef int v4i32 __attribute__((ext_vector_type(4)));
// buffer descriptor (raw, stride 0) over `bytes` bytes at p; every field is made wave-uniform
__device__ __forceinline__ v4i32 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  v4i32 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// One LDS-DMA piece: 64 lanes x 16 bytes from rs[voff] to LDS bytes [lds_base, lds_base + 1024).
// Issued from inline asm so the compiler does not serialise the following ds_reads behind it;
// completion is counted by hand (s_waitcnt vmcnt(0) before the barrier that publishes the tile).
__device__ __forceinline__ void lds_dma16(v4i32 rs, unsigned lds_base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(rs) : "memory");
}


// the same with a scalar byte offset added to every lane's (the k offset of a k-tile: no per-lane add per request)
__device__ __forceinline__ void lds_dma16_s(v4i32 rs, unsigned lds_base, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
struct GroupArgs {
  int n, total;
  int start[IFSEG_GEMM_GROUP_MAX + 1];     // first (XCD-remapped) tile of problem i; start[n] = total
  GemmArgs p[IFSEG_GEMM_GROUP_MAX];
};

}  // namespace
