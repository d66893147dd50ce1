// Shared device helpers for the ifseg_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef unsigned short bf16_t;  // raw storage type

#include <stdio.h>
#include <stdlib.h>
#define IFSEG_CHECK_LAUNCH()                                                                   \
  do {                                                                                         \
    hipError_t e__ = hipGetLastError();                                                        \
    if (e__ != hipSuccess) {                                                                   \
      if (getenv("IFSEG_DEBUG")) {                                                             \
        int d__ = -1;                                                                          \
        (void)hipGetDevice(&d__);                                                              \
        fprintf(stderr, "[ifseg_hip] %s:%d launch failed: %s (%d), device %d\n", __FILE__,    \
                __LINE__, hipGetErrorString(e__), (int)e__, d__);                              \
      }                                                                                        \
      return (int)e__;                                                                         \
    }                                                                                          \
  } while (0)

// The laboratory gate of the host code (ifseg_amd/lab.py is the Python side): a measurement switch IFSEG_<NAME> is honoured only
// when IFSEG_LAB=1 is set as well; bench.py refuses to report a number under IFSEG_LAB=1 unless started with --lab.  Callers keep
// the result in a function-local static (one environment scan per process, not per launch).
static inline const char* ifseg_lab_env(const char* name) {
  const char* g = getenv("IFSEG_LAB");
  return (g && g[0] == '1' && g[1] == 0) ? getenv(name) : nullptr;
}

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }

// round-to-nearest-even float -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two floats -> packed bf16x2 (RNE) in one VALU op (gfx950 v_cvt_pk_bf16_f32)
// (as a native conversion, not inline asm: the compiler's wait-count pass does not protect an inline-asm DESTINATION
// against a still-pending ds_read into the same VGPR -- the late LDS data then overwrote freshly packed P values)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2_t;
  typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2_t;
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// erf(x/sqrt2) by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution); e = exp(-x^2/2) is the
// same exponential the Gaussian pdf of the GELU derivative needs, so backward costs one exp per element.
__device__ __forceinline__ float erf_as(float x, float e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);      // v_rcp_f32 (1 ulp); __frcp_rn is a ten-instruction IEEE divide
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * e;
  return x < 0.f ? -r : r;
}
union U128 {
  uint4 v;
  bf16x8 b;
  uint32_t w[4];
  bf16_t h[8];
};
union U64 {
  uint2 v;
  s16x4 s;
  uint32_t w[2];
  bf16_t h[4];
};

// ---- LDS tile images (16 KiB each: 128 x 64 bf16) -------------------------
// "KC" tile: [128 rows][64 k], k contiguous in global.  16-byte chunks are
// XOR-swizzled inside 256-byte lines (two rows per line) so that the 16 rows a
// ds_read_b128 lane group touches at one k-chunk land on 16 distinct slots.
__device__ __forceinline__ int kc_off(int r, int c /*0..7*/) {
  int line = r >> 1;
  int s = (((r & 1) << 3) | c) ^ (line & 15);
  return line * 256 + s * 16;
}
// "KS" tile: [64 k rows][128 cols], cols contiguous in global (the reduction
// index is the row).  64-byte granules XOR-swizzled by (k & 3) so the four k
// rows one ds_read_b64_tr_b16 touches sit on four different bank quarters.
__device__ __forceinline__ int ks_off(int kr, int col /*0..127*/) {
  int c = col >> 3;
  return kr * 256 + ((((c >> 2) ^ (kr & 3))) << 6) + ((c & 3) << 4) + ((col & 7) << 1);
}

__device__ __forceinline__ bf16x8 lds_read_b128(const unsigned char* p) {
  U128 u;
  u.v = *reinterpret_cast<const uint4*>(p);
  return u.b;
}
// transposed 4x(16 lanes) read: lane i of a 16-lane group passes the address of
// chunk i (row i>>2, 4-col chunk i&3) of a [4][16] bf16 block and receives
// column i (4 rows).
__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(p));
}

// MFMA fragment (8 bf16 along k) for rows rb..rb+31 of a KC tile at k-step ks (16 k each)
__device__ __forceinline__ bf16x8 frag_kc(const unsigned char* tile, int rb, int ks, int lane) {
  return lds_read_b128(tile + kc_off(rb + (lane & 31), ks * 2 + (lane >> 5)));
}
// same fragment for columns cb..cb+31 of a KS tile
__device__ __forceinline__ bf16x8 frag_ks(const unsigned char* tile, int cb, int ks, int lane) {
  int i = lane & 15, g = (lane >> 4) & 1, kh = lane >> 5;
  int col = cb + g * 16 + (i & 3) * 4;
  int kr = ks * 16 + kh * 8 + (i >> 2);
  U64 a, b;
  a.s = lds_read_tr(tile + ks_off(kr, col));
  b.s = lds_read_tr(tile + ks_off(kr + 4, col));
  U128 u;
  u.w[0] = a.w[0]; u.w[1] = a.w[1]; u.w[2] = b.w[0]; u.w[3] = b.w[1];
  return u.b;
}

// ---- LDS-DMA (global -> LDS without staging registers) -----------------------
// One wave instruction moves 64 x 16 (or 64 x 4) bytes: lane l's source is its own
// address, its destination is lds_base + 16 l (4 l); lds_base must be wave-uniform.
// Issued from inline asm, so the compiler neither counts it nor orders ds_reads behind
// it: the issuing wave waits (lds_dma_wait) and then a barrier publishes the data.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ __forceinline__ void lds_dma16_g(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ __forceinline__ void lds_dma4_g(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
// (s_nop 4: the instruction reads SGPRs -- its base / resource descriptor -- that the compiler may have just reloaded from
// a spill lane with v_readlane; "VALU writes SGPR -> VMEM reads it" needs 5 wait states on gfx9 and the hazard
// recognizer cannot see into inline asm.  Observed as garbage dO tiles once the dK/dV kernel spilled a base.)
// the same with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit per-lane pointers live
__device__ __forceinline__ void lds_dma16_gs(const void* sbase, int voff_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff_bytes), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ __forceinline__ void lds_dma4_gs(const void* sbase, int voff_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1"
               :: "v"(voff_bytes), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Wave-wide sum, the same value in every lane.  DPP row shifts / row broadcasts feed the adds directly (six VALU
// instructions and one v_readlane; the butterfly of ds_bpermute it replaces costs six dependent LDS round trips,
// ~40 % of a LayerNorm-backward row's critical path at three waves per SIMD).  Inactive lanes contribute nothing, so
// call it from wave-uniform control flow.
__device__ __forceinline__ float warp_sum(float v) {
  auto shr = [](float x, auto ctrl_tag, auto rowmask_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value, RM = decltype(rowmask_tag)::value;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, RM, 0xf, true));
  };
  using std::integral_constant;
  v += shr(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});   // row_shr:1
  v += shr(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});   // row_shr:2
  v += shr(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});   // row_shr:4
  v += shr(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});   // row_shr:8  -> lane 15 of a row: row total
  v += shr(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
  v += shr(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3 -> lane 63: total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float warp_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// bijective XCD-aware remap of a linear block id: XCD x (= id % 8 as observed)
// gets one contiguous chunk of the tile space so neighbours share L2 panels.
// Causal attention: workgroups differ in length (a tile sees only part of the other sequence).  Launch order = tile index
// slowest (longest first when the caller maps rank 0 to its heaviest tile), (batch, head) fastest, and every (batch, head)
// stays on one XCD when their number divides by 8: -> (rank of the tile, batch*head index)
__device__ __forceinline__ void causal_order(int bid, int nbh, int* rank, int* bh) {
  if ((nbh & 7) == 0) {
    const int per = nbh >> 3, xcd = bid & 7, loc = bid >> 3;
    *rank = loc / per;
    *bh = xcd * per + loc % per;
  } else {
    *rank = bid / nbh;
    *bh = bid % nbh;
  }
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}
