// bf16 MFMA GEMM for gfx950 with fused epilogue, three operand layouts and an
// implicit-GEMM (NHWC gather) A loader for the frozen ResNet trunk.
//
//   NT : C[M,N] = A[M,K] . B[N,K]^T         forward Linear (reference: F.linear in
//        unify_multihead_attention.py:327-346,513; unify_transformer_layer.py:279-283)
//   NN : C[M,N] = A[M,K] . B[K,N]           dX = dY . W
//   TN : C[M,N] = A[K,M]^T . B[K,N]         dW = dY^T . X   (reduction over tokens)
//   CONV: A rows gathered from an NHWC image (resnet.py:117-137 convs, BN folded)
//
// 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16.  Operands are staged global -> registers -> LDS in
// their natural (coalesced) orientation; k-contiguous tiles are read with
// ds_read_b128, k-strided tiles with ds_read_b64_tr_b16 (hardware transpose), so
// no operand is ever transposed in HBM.  The MFMA is issued "swapped"
// (D[n][m]) so each lane owns 4 consecutive output columns -> 8/16-byte stores.
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr int BM = 128, BK = 64;   // BN is a template parameter (128, or 64 for narrow / few-tile outputs)
enum { A_KC = 0, A_KS = 1, A_CONV = 2 };

struct GemmArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  int M, N, K, lda, ldb, ldc;
  const bf16_t* bias; const bf16_t* resid; int ldr;
  float alpha; int alpha_ncols; int flags;
  int cH, cW, cC, cKW, cStride, cPad, cOH, cOW;
  long long sA, sB, sC, sR;
  int splitk, kchunk; long long sCsplit;
};

template <int AMODE, bool B_KS, int BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  static_assert(BN == 128 || (BN == 64 && !B_KS), "64-wide tiles only for k-contiguous B");
  constexpr int NJ = BN / 64;        // 32-column MFMA tiles per wave along N
  constexpr int NBI = BN / 32;       // B staging chunks per thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 16384;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const bf16_t* Ab = g.A + (long long)blockIdx.y * g.sA;
  const bf16_t* Bb = g.B + (long long)blockIdx.y * g.sB;

  // ---- per-thread staging geometry -------------------------------------
  // KC tile: chunk (row = tid/8 + 32 i, c = tid%8); KS tile: (krow = tid/16 + 16 i, c = tid%16)
  uint4 ra[4], rb[4];
  long long a_rowbase[4];  // element offset of the row start (KC / CONV), or -1
  int cv_b[4], cv_oy[4], cv_ox[4];
  if (AMODE == A_KC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (tid >> 3) + 32 * i;
      a_rowbase[i] = (m < g.M) ? (long long)m * g.lda : -1;
    }
  } else if (AMODE == A_CONV) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (tid >> 3) + 32 * i;
      if (m < g.M) {
        cv_ox[i] = m % g.cOW;
        int q = m / g.cOW;
        cv_oy[i] = q % g.cOH;
        cv_b[i] = q / g.cOH;
      } else {
        cv_b[i] = -1; cv_oy[i] = 0; cv_ox[i] = 0;
      }
    }
  }
  long long b_rowbase[4];
  if (!B_KS) {
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      int n = n0 + (tid >> 3) + 32 * i;
      b_rowbase[i] = (n < g.N) ? (long long)n * g.ldb : -1;
    }
  }

  auto load_tiles = [&](int k0) {
    // ---- A ----
    if (AMODE == A_KC) {
      const int kc = k0 + (tid & 7) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = make_uint4(0, 0, 0, 0);
        if (a_rowbase[i] >= 0 && kc < g.K)
          ra[i] = *reinterpret_cast<const uint4*>(Ab + a_rowbase[i] + kc);
      }
    } else if (AMODE == A_CONV) {
      const int tap = k0 / g.cC, c0 = k0 - tap * g.cC + (tid & 7) * 8;
      const int ky = tap / g.cKW, kx = tap - ky * g.cKW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = make_uint4(0, 0, 0, 0);
        int iy = cv_oy[i] * g.cStride + ky - g.cPad, ix = cv_ox[i] * g.cStride + kx - g.cPad;
        if (cv_b[i] >= 0 && k0 < g.K && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW)
          ra[i] = *reinterpret_cast<const uint4*>(
              Ab + (((long long)cv_b[i] * g.cH + iy) * g.cW + ix) * g.cC + c0);
      }
    } else {  // A_KS: global [K][M], tile [64 k][128 m]
      const int mc = m0 + (tid & 15) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int k = k0 + (tid >> 4) + 16 * i;
        ra[i] = make_uint4(0, 0, 0, 0);
        if (k < g.K && mc < g.M) ra[i] = *reinterpret_cast<const uint4*>(Ab + (long long)k * g.lda + mc);
      }
    }
    // ---- B ----
    if (!B_KS) {
      const int kc = k0 + (tid & 7) * 8;
#pragma unroll
      for (int i = 0; i < NBI; ++i) {
        rb[i] = make_uint4(0, 0, 0, 0);
        if (b_rowbase[i] >= 0 && kc < g.K)
          rb[i] = *reinterpret_cast<const uint4*>(Bb + b_rowbase[i] + kc);
      }
    } else {
      const int nc = n0 + (tid & 15) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int k = k0 + (tid >> 4) + 16 * i;
        rb[i] = make_uint4(0, 0, 0, 0);
        if (k < g.K && nc < g.N) rb[i] = *reinterpret_cast<const uint4*>(Bb + (long long)k * g.ldb + nc);
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AMODE == A_KS)
        *reinterpret_cast<uint4*>(sA + ks_off((tid >> 4) + 16 * i, (tid & 15) * 8)) = ra[i];
      else
        *reinterpret_cast<uint4*>(sA + kc_off((tid >> 3) + 32 * i, tid & 7)) = ra[i];
      if (B_KS)
        *reinterpret_cast<uint4*>(sB + ks_off((tid >> 4) + 16 * i, (tid & 15) * 8)) = rb[i];
      else if (i < NBI)
        *reinterpret_cast<uint4*>(sB + kc_off((tid >> 3) + 32 * i, tid & 7)) = rb[i];
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // split-K: slice blockIdx.z reduces k in [kbeg, kend) into its own fp32 slab of C
  const int kbeg = g.splitk > 1 ? blockIdx.z * g.kchunk : 0;
  const int kend = g.splitk > 1 ? min(g.K, kbeg + g.kchunk) : g.K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (g.splitk > 1) g.K = kend;       // loaders zero-fill beyond the slice
  load_tiles(kbeg);
  store_tiles();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);  // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[2], fb[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = (AMODE == A_KS) ? frag_ks(sA, wm * 64 + i * 32, ks, lane) : frag_kc(sA, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        fb[j] = B_KS ? frag_ks(sB, wn * (BN / 2) + j * 32, ks, lane) : frag_kc(sB, wn * (BN / 2) + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tiles();
      __syncthreads();
    }
  }

  // ---- epilogue: lane owns row m = ..+(lane&31), 4 consecutive columns per reg group
  const bool relu = g.flags & IFSEG_GEMM_RELU, out_f32 = g.flags & IFSEG_GEMM_OUT_F32,
             accum = g.flags & IFSEG_GEMM_ACCUMULATE;
  const bf16_t* Rb = g.resid ? g.resid + (long long)blockIdx.y * g.sR : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * (BN / 2) + j * 32 + 8 * rg + 4 * (lane >> 5);
        if (n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
        if (g.bias) {
          uint2 bw = *reinterpret_cast<const uint2*>(g.bias + n);
          v[0] += bflo(bw.x); v[1] += bfhi(bw.x); v[2] += bflo(bw.y); v[3] += bfhi(bw.y);
        }
        if (n < g.alpha_ncols) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= g.alpha;
        }
        if (Rb) {
          uint2 rw = *reinterpret_cast<const uint2*>(Rb + (long long)m * g.ldr + n);
          v[0] += bflo(rw.x); v[1] += bfhi(rw.x); v[2] += bflo(rw.y); v[3] += bfhi(rw.y);
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (out_f32) {
          float* cp = reinterpret_cast<float*>(g.C) + (long long)blockIdx.y * g.sC + (long long)blockIdx.z * g.sCsplit +
                      (long long)m * g.ldc + n;
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (accum) {
            float4 p = *reinterpret_cast<float4*>(cp);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
          }
          *reinterpret_cast<float4*>(cp) = o;
        } else {
          bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long long)blockIdx.y * g.sC + (long long)m * g.ldc + n;
          if (accum) {
            uint2 pw = *reinterpret_cast<const uint2*>(cp);
            v[0] += bflo(pw.x); v[1] += bfhi(pw.x); v[2] += bflo(pw.y); v[3] += bfhi(pw.y);
          }
          *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
      }
    }
  }
}

// skinny projection: C[M, N<=32*?] = A[M,K] . B[N,K]^T for tiny N (seg tokens, 15..171):
// one wave per row block is overkill -- handled by the generic kernel with N padded
// by the caller when N % 8 == 0; otherwise see ifseg_rowdot in rowops.hip.

}  // namespace

extern "C" int ifseg_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K,
                               int lda, int ldb, int ldc, const void* bias, float alpha, int alpha_ncols,
                               const void* resid, int ldr, int flags, int batch, long long strideA,
                               long long strideB, long long strideC, long long strideR, int splitk, void* stream) {
  (void)hipGetLastError();
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 3) || (resid && (ldr & 3))) return IFSEG_ERR_BAD_SHAPE;
  if (layout == IFSEG_GEMM_NT && (K & 7)) return IFSEG_ERR_BAD_SHAPE;
  if (layout == IFSEG_GEMM_NN && (K & 7)) return IFSEG_ERR_BAD_SHAPE;
  if (layout == IFSEG_GEMM_TN && (M & 7)) return IFSEG_ERR_BAD_SHAPE;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.bias = (const bf16_t*)bias; g.resid = (const bf16_t*)resid; g.ldr = ldr;
  g.alpha = alpha; g.alpha_ncols = (alpha == 1.0f) ? 0 : (alpha_ncols < 0 ? N : alpha_ncols);
  g.flags = flags; g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sR = strideR;
  g.splitk = 1;
  if (splitk > 1) {
    // C must be an fp32 workspace [splitk][M][ldc]; epilogue extras are not applied to partial sums
    if (!(flags & IFSEG_GEMM_OUT_F32) || (flags & IFSEG_GEMM_ACCUMULATE) || bias || resid || alpha != 1.0f)
      return IFSEG_ERR_BAD_ARG;
    g.kchunk = (((K + splitk - 1) / splitk) + BK - 1) / BK * BK;
    g.splitk = (K + g.kchunk - 1) / g.kchunk;
    g.sCsplit = (long long)M * ldc;
  }
  const int tiles128 = ((M + BM - 1) / BM) * ((N + 127) / 128);
  // narrow tiles when the output is narrow or there are too few 128-wide tiles to fill 256 CUs
  const bool narrow = layout == IFSEG_GEMM_NT && g.splitk == 1 && (N <= 64 || tiles128 < 384);
  const int tiles = narrow ? ((M + BM - 1) / BM) * ((N + 63) / 64) : tiles128;
  dim3 grid(tiles, batch > 0 ? batch : 1, g.splitk), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (layout < 0 || layout > 2) return IFSEG_ERR_BAD_ARG;
  const double nb = batch > 0 ? batch : 1;
  ifseg_prof_begin(IFSEG_K_GEMM_NT + layout, s, 2.0 * M * N * K * nb, 2.0 * nb * ((double)M * K + (double)N * K + (double)M * N));
  switch (layout) {
    case IFSEG_GEMM_NT:
      if (narrow) hipLaunchKernelGGL((gemm_kernel<A_KC, false, 64>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_kernel<A_KC, false, 128>), grid, block, 0, s, g);
      break;
    case IFSEG_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<A_KC, true, 128>), grid, block, 0, s, g); break;
    case IFSEG_GEMM_TN: hipLaunchKernelGGL((gemm_kernel<A_KS, true, 128>), grid, block, 0, s, g); break;
  }
  ifseg_prof_end(IFSEG_K_GEMM_NT + layout, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// Implicit-GEMM convolution on an NHWC bf16 image with folded FrozenBN:
//   out[b,oy,ox,co] = act( sum_{ky,kx,ci} in[b, oy*s+ky-p, ox*s+kx-p, ci] * w[co,ky,kx,ci]
//                          + shift[co] + resid[b,oy,ox,co] )
// (reference: Bottleneck.forward resnet.py:117-137, FrozenBatchNorm2d frozen_bn.py:36-57)
extern "C" int ifseg_conv2d_nhwc_bf16(const void* in, const void* w, const void* shift, const void* resid,
                                      void* out, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad, int relu, void* stream) {
  (void)hipGetLastError();
  if ((Cin % 64) || (Cout & 7)) return IFSEG_ERR_BAD_SHAPE;
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  GemmArgs g{};
  g.A = (const bf16_t*)in; g.B = (const bf16_t*)w; g.C = out;
  g.M = B * OH * OW; g.N = Cout; g.K = KH * KW * Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.bias = (const bf16_t*)shift; g.resid = (const bf16_t*)resid; g.ldr = Cout;
  g.alpha = 1.f; g.alpha_ncols = 0; g.flags = relu ? IFSEG_GEMM_RELU : 0;
  g.cH = H; g.cW = W; g.cC = Cin; g.cKW = KW; g.cStride = stride; g.cPad = pad; g.cOH = OH; g.cOW = OW;
  const int tiles128 = ((g.M + BM - 1) / BM) * ((g.N + 127) / 128);
  const bool narrow = g.N <= 64 || tiles128 < 384;
  const int tiles = narrow ? ((g.M + BM - 1) / BM) * ((g.N + 63) / 64) : tiles128;
  ifseg_prof_begin(IFSEG_K_CONV, (hipStream_t)stream, 2.0 * g.M * g.N * g.K, 2.0 * ((double)B * H * W * Cin + (double)g.N * g.K + (double)g.M * g.N));
  if (narrow) hipLaunchKernelGGL((gemm_kernel<A_CONV, false, 64>), dim3(tiles, 1), dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((gemm_kernel<A_CONV, false, 128>), dim3(tiles, 1), dim3(256), 0, (hipStream_t)stream, g);
  ifseg_prof_end(IFSEG_K_CONV, (hipStream_t)stream);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_abi_version(void) { return IFSEG_ABI_VERSION; }
