// bf16 MFMA GEMM for gfx950 with fused epilogue, three operand layouts and an
// implicit-GEMM (NHWC gather) A loader for the frozen ResNet trunk.
//
//   NT : C[M,N] = A[M,K] . B[N,K]^T         forward Linear (reference: F.linear in
//        unify_multihead_attention.py:327-346,513; unify_transformer_layer.py:279-283)
//   NN : C[M,N] = A[M,K] . B[K,N]           dX = dY . W
//   TN : C[M,N] = A[K,M]^T . B[K,N]         dW = dY^T . X   (reduction over tokens)
//   CONV: A rows gathered from an NHWC image (resnet.py:117-137 convs, BN folded)
//
// 128x128x64 block tile (128x64 for narrow outputs), 4 waves (2x2), each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16.  Operand tiles go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
// registers, zero padding from the buffer descriptor) in their natural (coalesced) orientation, one or two
// LDS stages; k-contiguous tiles are read with ds_read_b128, k-strided tiles with ds_read_b64_tr_b16
// (hardware transpose), so no operand is ever transposed in HBM.  The MFMA is issued "swapped" (D[n][m]) so
// each lane owns 4 consecutive output columns -> 8/16-byte stores.  Split-K (weight gradients) writes fp32
// slabs that a second launch sums: a fused "last workgroup reduces" was measured 2-3x slower on MI355X -- the
// agent-scope release/acquire it needs writes back / invalidates the per-XCD L2s.
#include <cstdlib>
#include "gemm_common.h"
#include "prof.h"

namespace {

// One output tile (and, with split-K, one k-slice of it).  bx / by / bz are the workgroup's tile, batch and k-slice
// coordinates (blockIdx of the plain kernel; the grouped kernel passes the tile index inside its problem);
// `remapped`: bx is already an XCD-remapped tile index.
template <int AMODE, bool B_KS, int BN, int BK, int STAGES, bool COLSUM = false, bool EPI_GLN = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz_in, const bool remapped) {
  static_assert(BN == 128 || (BN == 64 && !B_KS), "64-wide tiles only for k-contiguous B");
  constexpr int NJ = BN / 64;                 // 32-column MFMA tiles per wave along N
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int LA = A_BYTES / 4096, LB = B_BYTES / 4096;   // 1-KiB DMA pieces per wave
  constexpr int KS_STEPS = BK / 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  // split-K with xcd_groups > 0 (1-D grid of tiles x slices): XCD x = workgroup id & 7 owns k-slice x / groups and the
  // contiguous tile range x % groups of it, so the rows of A and B a slice touches stream through ONE XCD's L2 instead of
  // being fetched from HBM by every XCD that happens to hold a few of its tiles (measured 3.6-5.7x over-fetch)
  int t, bz = bz_in;
  if (g.xcd_groups > 0) {
    const int xcd = bx & 7, loc = bx >> 3;
    bz = xcd / g.xcd_groups;
    t = (xcd % g.xcd_groups) * ((tiles_m * tiles_n) / g.xcd_groups) + loc;
  } else {
    t = remapped ? bx : xcd_remap(bx, tiles_m * tiles_n);
  }
  // tile order: column tiles fastest -- or (grouped weight gradients of a problem with fewer row than column tiles, fc2's
  // 6 x 24) row tiles fastest: the XCD's run of consecutive tiles then shares the FEW panels of the short side and streams
  // only its own part of the long side, instead of every XCD reading all of the long operand
  // (round 6: -0.02 ... -0.08 ms per step, profiles/round6_dw_small_experiments.txt)
  const bool mfast = COLSUM && (g.flags & GEMM_FLAG_MFAST) && tiles_m < tiles_n;
  const int m0 = (mfast ? t % tiles_m : t / tiles_n) * BM, n0 = (mfast ? t / tiles_m : t % tiles_n) * BN;
  const v4i32 rsA = make_rsrc(g.A + (long long)by * g.sA, g.nrecA);
  const v4i32 rsB = make_rsrc(g.B + (long long)by * g.sB, g.nrecB);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // split-K: slice bz reduces k in [kbeg, kend) into its own fp32 slab of C
  const int kbeg = g.splitk > 1 ? bz * g.kchunk : 0;
  const int kend = g.splitk > 1 ? min(g.K, kbeg + g.kchunk) : g.K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  const bool ktail = (kend - kbeg) & (BK - 1);

  // ---- per-lane source offsets (bytes, at k = kbeg); rows outside the matrix -> OOB (zeros)
  unsigned offA[LA], offB[LB];
  int c8A[LA], c8B[LB];                 // k offset of the lane's chunk inside a KC tile (k-tail test)
  // conv: per-lane byte offset of tap (0,0) of the lane's output pixel (may lie in the padding: the
  // sum with the tap offset is only used when the tap is in bounds) and the pixel's top-left input coords
  int cv_base[LA], cv_iy0[LA], cv_ix0[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    if (AMODE == A_KS) {
      int kr, col;
      ks_src(wave * LA + i, lane, kr, col);
      offA[i] = (m0 + col < g.M) ? (unsigned)(((long long)(kbeg + kr) * g.lda + m0 + col) * 2) : OOB;
      c8A[i] = 0;
    } else {
      int row, c;
      kct_src<BK>(wave * LA + i, lane, row, c);
      const int m = m0 + row;
      c8A[i] = c * 8;
      if (AMODE == A_KC) {
        offA[i] = (m < g.M) ? (unsigned)(((long long)m * g.lda + kbeg + c * 8) * 2) : OOB;
      } else {
        offA[i] = 0;
        if (m < g.M) {
          const int ox = m % g.cOW, q = m / g.cOW, oy = q % g.cOH, b = q / g.cOH;
          cv_iy0[i] = oy * g.cStride - g.cPad;
          cv_ix0[i] = ox * g.cStride - g.cPad;
          cv_base[i] = (((b * g.cH + cv_iy0[i]) * g.cW + cv_ix0[i]) * g.cC + c * 8) * 2;
        } else {
          cv_iy0[i] = -0x40000000; cv_ix0[i] = -0x40000000; cv_base[i] = 0;   // never in bounds
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    if (B_KS) {
      int kr, col;
      ks_src(wave * LB + i, lane, kr, col);
      offB[i] = (n0 + col < g.N) ? (unsigned)(((long long)(kbeg + kr) * g.ldb + n0 + col) * 2) : OOB;
      c8B[i] = 0;
    } else {
      int row, c;
      kct_src<BK>(wave * LB + i, lane, row, c);
      const int n = n0 + row;
      c8B[i] = c * 8;
      offB[i] = (n < g.N) ? (unsigned)(((long long)n * g.ldb + kbeg + c * 8) * 2) : OOB;
    }
  }
  // conv: (ky, kx, c0) of the next k-tile, advanced incrementally (tiles are issued in k order)
  int cky = 0, ckx = 0, cc0 = 0;
  if (AMODE == A_CONV) {
    const int tap = kbeg / g.cC;
    cc0 = kbeg - tap * g.cC;
    cky = tap / g.cKW;
    ckx = tap - cky * g.cKW;
  }
  const unsigned kadvA = (AMODE == A_KS) ? (unsigned)BK * g.lda * 2 : BK * 2;
  const unsigned kadvB = B_KS ? (unsigned)BK * g.ldb * 2 : BK * 2;

  auto issue = [&](int kt, int st) {
    const unsigned dA = lds0 + st * STAGE_BYTES + wave * (LA * 1024);
    const unsigned dB = lds0 + st * STAGE_BYTES + A_BYTES + wave * (LB * 1024);
    const int krem = (kend - kbeg) - kt * BK;           // valid k in this tile (>= BK except on a tail)
    const bool tail = ktail && kt == nk - 1;
    if (AMODE == A_CONV) {
      const int delta = ((cky * g.cW + ckx) * g.cC + cc0) * 2;       // wave-uniform byte offset of this tap / channel block
      const bool interior = g.cPad == 0 && g.cKW == 1;               // 1x1 convolutions never leave the image
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        unsigned v = (unsigned)(cv_base[i] + delta);
        const bool ok = interior ? cv_iy0[i] >= 0
                                 : ((unsigned)(cv_iy0[i] + cky) < (unsigned)g.cH && (unsigned)(cv_ix0[i] + ckx) < (unsigned)g.cW);
        lds_dma16(rsA, dA + i * 1024, ok ? v : OOB);
      }
      cc0 += BK;
      if (cc0 >= g.cC) { cc0 = 0; if (++ckx == g.cKW) { ckx = 0; ++cky; } }
    } else {
      // (a full k-tile: the k offset rides in the instruction's scalar offset -- no per-piece VALU; a k-tail takes its zero
      // padding from the per-lane offset)
      const unsigned ka = (unsigned)kt * kadvA;
#ifdef GEMM_NO_SOFF
      if (false) {
#else
      if (!tail) {
#endif
#pragma unroll
        for (int i = 0; i < LA; ++i) lds_dma16_s(rsA, dA + i * 1024, offA[i], ka);
      } else {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
          unsigned v = offA[i] + ka;
          if (AMODE == A_KC && c8A[i] >= krem) v = OOB;
          lds_dma16(rsA, dA + i * 1024, v);
        }
      }
    }
    const unsigned kb = (unsigned)kt * kadvB;
#ifdef GEMM_NO_SOFF
    if (false) {
#else
    if (!tail) {
#endif
#pragma unroll
      for (int i = 0; i < LB; ++i) lds_dma16_s(rsB, dB + i * 1024, offB[i], kb);
    } else {
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        unsigned v = offB[i] + kb;
        if (!B_KS && c8B[i] >= krem) v = OOB;
        lds_dma16(rsB, dB + i * 1024, v);
      }
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // COLSUM (weight-gradient GEMMs): the column sums of A (= the bias gradient) come out of the same A fragments
  // through one more MFMA against an all-ones B fragment, in the workgroups of the first column tile only
  f32x16 accb[COLSUM ? 2 : 1];
  const bool do_colsum = COLSUM && n0 == 0 && wn == 0 && (g.flags & IFSEG_GEMM_COLSUM);
  if (COLSUM) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
  }

  auto compute = [&](int st) {
    const unsigned char* sA = smem + st * STAGE_BYTES;
    const unsigned char* sB = sA + A_BYTES;
    auto ldfrag = [&](int ks, bf16x8* fa, bf16x8* fb) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = (AMODE == A_KS) ? frag_ks(sA, wm * 64 + i * 32, ks, lane) : frag_kct<BK>(sA, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        fb[j] = B_KS ? frag_ks(sB, wn * (BN / 2) + j * 32, ks, lane) : frag_kct<BK>(sB, wn * (BN / 2) + j * 32, ks, lane);
    };
    if constexpr (!COLSUM && AMODE == A_KC && B_KS) {
      // dX GEMMs (B = W read k-strided: two transposed LDS reads per fragment): the fragments of k-slice ks+1 are requested
      // before the MFMAs of slice ks (two register sets, 112 -> 124 VGPRs, still four waves per SIMD): -4..17 % on the
      // shapes of the model.  The same for the NT kernel costs its bias prefetch or a wave per SIMD and loses (measured).
      bf16x8 fa[2][2], fb[2][NJ];
      ldfrag(0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < KS_STEPS; ++ks) {
        if (ks + 1 < KS_STEPS) ldfrag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][j], fa[ks & 1][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < KS_STEPS; ++ks) {
      bf16x8 fa[2], fb[NJ];
      ldfrag(ks, fa, fb);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if constexpr (COLSUM) {
        if (do_colsum) {
          U128 one;
          one.w[0] = one.w[1] = one.w[2] = one.w[3] = 0x3F803F80u;      // eight bf16 1.0
#pragma unroll
          for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one.b, fa[i], accb[i], 0, 0, 0);
        }
      }
    }
    }
  };

  // the lane's bias values (4 consecutive columns per register group) are requested before the k-loop: loaded in the
  // epilogue their L2 round trip sits in the tail of every tile (~6 us of a 60 us K = 768 GEMM)
  constexpr bool PRE_BIAS = AMODE == A_KC && !B_KS;      // F.linear forward; elsewhere the 16 registers cost a wave per SIMD
  uint2 biasr[PRE_BIAS ? NJ : 1][4];
  if (PRE_BIAS && g.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * (BN / 2) + j * 32 + 8 * rg + 4 * (lane >> 5);
        biasr[j][rg] = (n < g.N) ? *reinterpret_cast<const uint2*>(g.bias + n) : make_uint2(0, 0);
      }
  }
  if constexpr (STAGES == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    // tile kt+1 streams into the other stage while tile kt feeds the MFMAs: one barrier per k-step
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      compute(kt & 1);
    }
  }

  if constexpr (COLSUM) {
    // every row of accb holds sum_k A[k][m]; lanes 0..31 carry row 0 in register 0.  Slab layout per k-slice:
    // [M x ldc | M] floats, so one reduction pass over M*ldc + M elements yields dW followed by db.
    if (do_colsum && lane < 32) {
      if (g.splitk > 1) {
        float* cb = reinterpret_cast<float*>(g.C) + (long long)bz * g.sCsplit + (long long)g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = m0 + wm * 64 + i * 32 + lane;
          if (m < g.M) cb[m] = accb[i][0];
        }
      } else {
        // no split: the sums are final -- db (bf16) sits right behind dW [M x ldc] in the gradient arena
        bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + (long long)g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = m0 + wm * 64 + i * 32 + lane;
          if (m < g.M) cb[m] = f2bf(accb[i][0] + ((g.flags & IFSEG_GEMM_ACCUMULATE) ? bf2f(cb[m]) : 0.f));
        }
      }
    }
  }
  // ---- epilogue: lane owns row m = ..+(lane&31), 4 consecutive columns per reg group
  const bool relu = g.flags & IFSEG_GEMM_RELU, out_f32 = g.flags & IFSEG_GEMM_OUT_F32,
             accum = g.flags & IFSEG_GEMM_ACCUMULATE;
  const bf16_t* Rb = g.resid ? g.resid + (long long)by * g.sR : nullptr;
  // bf16 output of a 128-wide tile leaves through LDS: a lane owns a ROW of the MFMA tile (4-column runs), so direct
  // stores write 32-byte pieces of 32 different rows per instruction (four requests per 128-byte line).  Each wave
  // parks its 64 x 64 sub-tile in its own 8 KiB of the (now idle) operand stage -- 16-byte chunks XOR-swizzled by the row
  // so neither the writes (8 consecutive rows per lane group) nor the reads conflict -- and writes it back as 8 full
  // 128-byte row segments per instruction.
  const bool lds_out = BN == 128 && !out_f32 && !accum && !((g.ldc | g.sC) & 7) && !((size_t)g.C & 15);
  unsigned char* sOut = smem + wave * 8192;
  if (BN == 128 && STAGES == 2) __syncthreads();      // (the one-stage loop ends with a barrier) operands are dead
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    const bool mvalid = m < g.M;
    if (!lds_out && !mvalid) continue;
    float dsum = 0.f;
    float gl_mu = 0.f, gl_rs = 0.f, gl_c1 = 0.f, gl_c2 = 0.f;
    if constexpr (EPI_GLN) {
      if (mvalid) { gl_mu = g.gln_mean[m]; gl_rs = g.gln_rstd[m]; gl_c1 = g.gln_c[2 * m]; gl_c2 = g.gln_c[2 * m + 1]; }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nj = n0 + wn * (BN / 2) + j * 32;
      uint2 held[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = nj + 8 * rg + 4 * (lane >> 5);
        held[rg] = make_uint2(0, 0);
        if (n >= g.N || !mvalid) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
        if (g.bias) {
          const uint2 bw = PRE_BIAS ? biasr[PRE_BIAS ? j : 0][rg] : *reinterpret_cast<const uint2*>(g.bias + n);
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
        if constexpr (EPI_GLN) {
          const uint2 uw = *reinterpret_cast<const uint2*>(g.gln_u + (long long)m * g.gln_ldu + n);
          const float4 gm = *reinterpret_cast<const float4*>(g.gln_gamma + n);
          const float uu[4] = {bflo(uw.x), bfhi(uw.x), bflo(uw.y), bfhi(uw.y)};
          const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = uu[e], ex = __expf(-0.5f * xv * xv), er = erf_as(xv, ex);
            const float act = 0.5f * xv * (1.f + er);
            const float dact = 0.5f * (1.f + er) + xv * 0.39894228040143268f * ex;
            const float xh = (act - gl_mu) * gl_rs;
            v[e] = gl_rs * (gg[e] * v[e] - gl_c1 - xh * gl_c2) * dact;
          }
        }
        if (out_f32) {
          float* cp = reinterpret_cast<float*>(g.C) + (long long)by * g.sC + (long long)bz * g.sCsplit +
                      (long long)m * g.ldc + n;
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (accum) {
            float4 p = *reinterpret_cast<float4*>(cp);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
          }
          *reinterpret_cast<float4*>(cp) = o;
        } else {
          bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long long)by * g.sC + (long long)m * g.ldc + n;
          if (accum) {
            uint2 pw = *reinterpret_cast<const uint2*>(cp);
            v[0] += bflo(pw.x); v[1] += bfhi(pw.x); v[2] += bflo(pw.y); v[3] += bfhi(pw.y);
          }
          const uint2 ow = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
          if (lds_out) held[rg] = ow; else *reinterpret_cast<uint2*>(cp) = ow;
          if (g.dot) {
            const uint2 dw = *reinterpret_cast<const uint2*>(g.dot + (long long)m * g.ldd + n);
            dsum += bflo(ow.x) * bflo(dw.x) + bfhi(ow.x) * bfhi(dw.x) + bflo(ow.y) * bflo(dw.y) + bfhi(ow.y) * bfhi(dw.y);
          }
        }
      }
      if (lds_out) {
        // the two lanes of a row (l, l + 32) hold alternating 4-column runs: they exchange one run each, so that a lane
        // holds 8 consecutive columns = one 16-byte chunk (chunk index j*4 + rgp*2 + half of the wave's 64-column row)
        const int R = i * 32 + (lane & 31);
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {
          const auto p0 = __builtin_amdgcn_permlane32_swap(held[2 * rgp].x, held[2 * rgp + 1].x, false, false);
          const auto p1 = __builtin_amdgcn_permlane32_swap(held[2 * rgp].y, held[2 * rgp + 1].y, false, false);
          const int c = j * 4 + rgp * 2 + (lane >> 5);
          *reinterpret_cast<uint4*>(sOut + R * 128 + ((c ^ (R & 7)) << 4)) = make_uint4(p0[0], p1[0], p0[1], p1[1]);
        }
      }
    }
    if (g.dot && mvalid) {
      // the wave's 64 columns are one head; lanes l and l + 32 hold the two interleaved halves of row m
      dsum += __shfl_xor(dsum, 32);
      const int hd = (n0 + wn * (BN / 2)) >> 6;
      if (lane < 32 && (hd << 6) < g.N)
        g.dot_out[((long long)(m / g.dot_T) * (g.N >> 6) + hd) * g.dot_T + (m % g.dot_T)] = dsum;
    }
  }
  if (lds_out) {
    const int c = lane & 7, nc = n0 + wn * 64 + c * 8;
    bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + (long long)by * g.sC + nc;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int R = it * 8 + (lane >> 3), m = m0 + wm * 64 + R;
      const uint4 v = *reinterpret_cast<const uint4*>(sOut + R * 128 + ((c ^ (R & 7)) << 4));
      if (m < g.M && nc + 8 <= g.N) *reinterpret_cast<uint4*>(cb + (long long)m * g.ldc) = v;
    }
  }
}

template <int AMODE, bool B_KS, int BN, int BK, int STAGES, bool COLSUM = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  gemm_tile<AMODE, B_KS, BN, BK, STAGES, COLSUM>(g, blockIdx.x, blockIdx.y, blockIdx.z, false);
}
// dX GEMM with the "GELU + LayerNorm backward" epilogue (its own instantiation: the epilogue's registers must not cost
// the plain dX GEMMs their fourth wave per SIMD)
template <int STAGES>
__global__ __launch_bounds__(256, 2) void gemm_nn_gln_kernel(GemmArgs g) {
  gemm_tile<A_KC, true, 128, GBK, STAGES, false, true>(g, blockIdx.x, blockIdx.y, blockIdx.z, false);
}

// Grouped weight-gradient GEMM: up to IFSEG_GEMM_GROUP_MAX independent TN problems (the dW = dY^T X products of one
// transformer layer's Linear modules) in ONE launch, one workgroup per 128x128 output tile walking ALL tokens -- no
// split-K, no fp32 slabs, no reduction launches; db rides on the same pass (COLSUM) and dW / db are written once, as
// bf16, straight into the gradient arena.  A single dW product of SegOFA-Base has 36..144 tiles (it cannot fill 256
// CUs without split-K); a layer's products together have 432 (encoder) / 576 (decoder).
// The grid is capped (gridDim.x <= total, a multiple of 8): a workgroup walks tiles blockIdx.x, + gridDim.x, ... .  The
// launch runs on the weight-gradient stream NEXT to the dX chain: with one workgroup per CU it leaves half of every
// CU's LDS and wave slots to the main stream's kernels -- launched with one workgroup per tile, its long-running
// workgroups (133 k-steps each) filled every CU and the main queue stalled 100-180 us per layer behind them.
template <int STAGES>
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(GroupArgs ga) {
  for (int w = blockIdx.x; w < ga.total; w += gridDim.x) {
    // consecutive remapped ids share an XCD (its L2 then holds the panels neighbouring tiles of one problem share)
    const int id = xcd_remap(w, ga.total);
    int pid = 0;
    for (int i = 1; i < ga.n; ++i) pid = (id >= ga.start[i]) ? i : pid;
    gemm_tile<A_KS, true, 128, GBK, STAGES, true>(ga.p[pid], id - ga.start[pid], 0, 0, true);
    __syncthreads();          // the next tile refills the LDS stages
  }
}

// skinny projection: C[M, N<=32*?] = A[M,K] . B[N,K]^T for tiny N (seg tokens, 15..171):
// one wave per row block is overkill -- handled by the generic kernel with N padded
// by the caller when N % 8 == 0; otherwise see ifseg_rowdot in rowops.hip.

}  // namespace

static int two_stage_max() {
  static int v = -1;
  if (v < 0) { const char* e = ifseg_lab_env("IFSEG_GEMM_TWO_STAGE_MAX"); v = e ? atoi(e) : TWO_STAGE_MAX_WGS; }
  return v;
}

static int gemm_impl(int layout, const void* A, const void* B, void* C, int M, int N, int K,
                     int lda, int ldb, int ldc, const void* bias, float alpha, int alpha_ncols,
                     const void* resid, int ldr, int flags, int batch, long long strideA,
                     long long strideB, long long strideC, long long strideR, int splitk, void* stream,
                     const void* dot, int ldd, float* dot_out, int dot_T, int prof_kind = -1) {
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
  if (dot) {
    // plain bf16 output, 128-wide tiles (a wave = 64 columns = one head), one batch
    if (layout != IFSEG_GEMM_NN || (N & 63) || (ldd & 3) || !dot_out || dot_T <= 0 || batch > 1 || splitk > 1 ||
        (flags & (IFSEG_GEMM_OUT_F32 | IFSEG_GEMM_COLSUM)))
      return IFSEG_ERR_BAD_ARG;
    g.dot = (const bf16_t*)dot; g.ldd = ldd; g.dot_out = dot_out; g.dot_T = dot_T;
  }
  g.splitk = 1;
  if (splitk > 1) {
    // C must be an fp32 workspace [splitk][M][ldc]; epilogue extras are not applied to partial sums
    if (!(flags & IFSEG_GEMM_OUT_F32) || (flags & IFSEG_GEMM_ACCUMULATE) || bias || resid || alpha != 1.0f)
      return IFSEG_ERR_BAD_ARG;
    g.kchunk = (((K + splitk - 1) / splitk) + 63) / 64 * 64;
    g.splitk = (K + g.kchunk - 1) / g.kchunk;
    g.sCsplit = (long long)M * ldc + ((flags & IFSEG_GEMM_COLSUM) ? ((M + 3) & ~3) : 0);
  }
  // bytes each buffer descriptor may address (loads beyond it return zeros: that is the k / row padding)
  const long long nrA = layout == IFSEG_GEMM_TN ? ((long long)(K - 1) * lda + M) * 2 : ((long long)(M - 1) * lda + K) * 2;
  const long long nrB = layout == IFSEG_GEMM_NT ? ((long long)(N - 1) * ldb + K) * 2 : ((long long)(K - 1) * ldb + N) * 2;
  if (nrA >= (1ll << 31) || nrB >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  g.nrecA = (unsigned)nrA; g.nrecB = (unsigned)nrB;
  const int tiles128 = ((M + BM - 1) / BM) * ((N + 127) / 128);
  // narrow tiles when the output is narrow or there are too few 128-wide tiles to fill 256 CUs
  const bool narrow = layout == IFSEG_GEMM_NT && g.splitk == 1 && (N <= 64 || tiles128 < 384);
  const int tiles = narrow ? ((M + BM - 1) / BM) * ((N + 63) / 64) : tiles128;
  dim3 grid(tiles, batch > 0 ? batch : 1, g.splitk), block(256);
  static const bool no_xcd_slices = ifseg_lab_env("IFSEG_GEMM_NO_XCD_SLICES") != nullptr;
  if (layout == IFSEG_GEMM_TN && g.splitk > 1 && batch <= 1 && (8 % g.splitk) == 0 && tiles % (8 / g.splitk) == 0 &&
      !no_xcd_slices) {
    g.xcd_groups = 8 / g.splitk;
    grid = dim3(tiles * g.splitk, 1, 1);
  }
  hipStream_t s = (hipStream_t)stream;
  if (layout < 0 || layout > 2) return IFSEG_ERR_BAD_ARG;
  if ((flags & IFSEG_GEMM_COLSUM) && layout != IFSEG_GEMM_TN) return IFSEG_ERR_BAD_ARG;
  const double nb = batch > 0 ? batch : 1;
  const int pk = prof_kind >= 0 ? prof_kind : IFSEG_K_GEMM_NT + layout;
  ifseg_prof_begin(pk, s, 2.0 * M * N * K * nb, 2.0 * nb * ((double)M * K + (double)N * K + (double)M * N));
  const bool two_stage = (long long)tiles * g.splitk * (batch > 0 ? batch : 1) <= two_stage_max();
#define LAUNCH2(AM, BKS, BNV)                                                                        \
  do {                                                                                               \
    if (two_stage) hipLaunchKernelGGL((gemm_kernel<AM, BKS, BNV, GBK, 2>), grid, block, 0, s, g);    \
    else hipLaunchKernelGGL((gemm_kernel<AM, BKS, BNV, GBK, 1>), grid, block, 0, s, g);              \
  } while (0)
  switch (layout) {
    case IFSEG_GEMM_NT:
      if (narrow) LAUNCH2(A_KC, false, 64);
      else LAUNCH2(A_KC, false, 128);
      break;
    case IFSEG_GEMM_NN: LAUNCH2(A_KC, true, 128); break;
    case IFSEG_GEMM_TN:
      if (flags & IFSEG_GEMM_COLSUM) {
        // split-K: fp32 slabs [M x ldc | M]; no split: bf16 dW with db written right behind it
        if (ldc != N || (splitk <= 1 && (flags & IFSEG_GEMM_OUT_F32)) || batch > 1) return IFSEG_ERR_BAD_ARG;
        if (two_stage) hipLaunchKernelGGL((gemm_kernel<A_KS, true, 128, GBK, 2, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel<A_KS, true, 128, GBK, 1, true>), grid, block, 0, s, g);
      } else {
        LAUNCH2(A_KS, true, 128);
      }
      break;
  }
  ifseg_prof_end(pk, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K,
                               int lda, int ldb, int ldc, const void* bias, float alpha, int alpha_ncols,
                               const void* resid, int ldr, int flags, int batch, long long strideA,
                               long long strideB, long long strideC, long long strideR, int splitk, void* stream) {
  return gemm_impl(layout, A, B, C, M, N, K, lda, ldb, ldc, bias, alpha, alpha_ncols, resid, ldr, flags, batch, strideA,
                   strideB, strideC, strideR, splitk, stream, nullptr, 0, nullptr, 0);
}

extern "C" int ifseg_gemm_tn_group(int n, const ifseg_gemm_tn_problem* probs, int max_workgroups, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  if (n > IFSEG_GEMM_GROUP_MAX || !probs) return IFSEG_ERR_BAD_ARG;
  GroupArgs ga{};
  ga.n = n;
  int total = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    const ifseg_gemm_tn_problem& q = probs[i];
    // C[M,N] (bf16, ldc == N, db[M] right behind it when `colsum`) = A[K,M]^T . B[K,N]
    if (q.M <= 0 || q.N <= 0 || q.K <= 0 || (q.M & 7) || (q.N & 7) || (q.lda & 7) || (q.ldb & 7) || !q.A || !q.B || !q.C)
      return IFSEG_ERR_BAD_SHAPE;
    GemmArgs& g = ga.p[i];
    g.A = (const bf16_t*)q.A; g.B = (const bf16_t*)q.B; g.C = q.C;
    g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.N;
    g.alpha = 1.f; g.splitk = 1;
    g.flags = (q.colsum ? IFSEG_GEMM_COLSUM : 0) | (q.accumulate ? IFSEG_GEMM_ACCUMULATE : 0) | GEMM_FLAG_MFAST;
    const long long nrA = ((long long)(q.K - 1) * q.lda + q.M) * 2, nrB = ((long long)(q.K - 1) * q.ldb + q.N) * 2;
    if (nrA >= (1ll << 31) || nrB >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
    g.nrecA = (unsigned)nrA; g.nrecB = (unsigned)nrB;
    ga.start[i] = total;
    total += ((q.M + BM - 1) / BM) * ((q.N + 127) / 128);
    flops += 2.0 * q.M * q.N * q.K;
    bytes += 2.0 * ((double)q.M * q.K + (double)q.N * q.K + (double)q.M * q.N);
  }
  ga.start[n] = total;
  ga.total = total;
  hipStream_t s = (hipStream_t)stream;
  ifseg_prof_begin(IFSEG_K_GEMM_TN, s, flops, bytes);
  int grid = total;
  if (max_workgroups > 0 && max_workgroups < total) grid = max_workgroups >= 8 ? (max_workgroups & ~7) : max_workgroups;
  // (two LDS stages; one stage -- half the footprint, four workgroups per CU -- measured 17.46 against 16.96 ms per step, round 5)
  hipLaunchKernelGGL(gemm_tn_group_kernel<2>, dim3(grid), dim3(256), 0, s, ga);
  ifseg_prof_end(IFSEG_K_GEMM_TN, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_gemm_nn_rowdot(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                    const void* dot, int ldd, float* dot_out, int rows_per_batch, void* stream) {
  return gemm_impl(IFSEG_GEMM_NN, A, B, C, M, N, K, lda, ldb, ldc, nullptr, 1.f, 0, nullptr, 0, 0, 1, 0, 0, 0, 0, 1, stream,
                   dot, ldd, dot_out, rows_per_batch);
}

extern "C" int ifseg_gemm_nn_gelu_ln_bwd(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                         const void* u, int ldu, const float* gamma, const float* mean, const float* rstd,
                                         const float* cstats, void* stream) {
  (void)hipGetLastError();
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (!A || !B || !C || !u || !gamma || !mean || !rstd || !cstats) return IFSEG_ERR_BAD_ARG;
  if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || (ldc & 3) || (ldu & 3) || ((size_t)gamma & 15)) return IFSEG_ERR_BAD_SHAPE;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.f; g.splitk = 1;
  g.gln_u = (const bf16_t*)u; g.gln_ldu = ldu; g.gln_gamma = gamma; g.gln_mean = mean; g.gln_rstd = rstd; g.gln_c = cstats;
  const long long nrA = ((long long)(M - 1) * lda + K) * 2, nrB = ((long long)(K - 1) * ldb + N) * 2;
  if (nrA >= (1ll << 31) || nrB >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  g.nrecA = (unsigned)nrA; g.nrecB = (unsigned)nrB;
  const int tiles = ((M + BM - 1) / BM) * ((N + 127) / 128);
  hipStream_t s = (hipStream_t)stream;
  // (timed with the NN family; flops of the GEMM only)
  ifseg_prof_begin(IFSEG_K_GEMM_NT + IFSEG_GEMM_NN, s, 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + 2.0 * M * N));
  if (tiles <= two_stage_max()) hipLaunchKernelGGL(gemm_nn_gln_kernel<2>, dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_nn_gln_kernel<1>, dim3(tiles), dim3(256), 0, s, g);
  ifseg_prof_end(IFSEG_K_GEMM_NT + IFSEG_GEMM_NN, s);
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
  if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {
    // a 1x1 stride-1 convolution on an NHWC image IS the F.linear product out[B H W, Cout] = in[B H W, Cin] . w[Cout, Cin]^T: the
    // plain k-contiguous loader (k offset in the DMA's scalar offset, no per-piece bounds test or tap arithmetic), same tiles, same
    // epilogue, same bits -- two of the three convolutions of every bottleneck (resnet.py:117-137).  Round 6.
    return gemm_impl(IFSEG_GEMM_NT, in, w, out, B * H * W, Cout, Cin, Cin, Cin, Cout, shift, 1.f, 0, resid, Cout,
                     relu ? IFSEG_GEMM_RELU : 0, 1, 0, 0, 0, 0, 1, stream, nullptr, 0, nullptr, 0, IFSEG_K_CONV);
  }
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  GemmArgs g{};
  g.A = (const bf16_t*)in; g.B = (const bf16_t*)w; g.C = out;
  g.M = B * OH * OW; g.N = Cout; g.K = KH * KW * Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.bias = (const bf16_t*)shift; g.resid = (const bf16_t*)resid; g.ldr = Cout;
  g.alpha = 1.f; g.alpha_ncols = 0; g.flags = relu ? IFSEG_GEMM_RELU : 0;
  g.cH = H; g.cW = W; g.cC = Cin; g.cKW = KW; g.cStride = stride; g.cPad = pad; g.cOH = OH; g.cOW = OW;
  const long long nrA = (long long)B * H * W * Cin * 2, nrB = (long long)g.N * g.K * 2;
  if (nrA >= (1ll << 31) || nrB >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  g.nrecA = (unsigned)nrA; g.nrecB = (unsigned)nrB;
  const int tiles128 = ((g.M + BM - 1) / BM) * ((g.N + 127) / 128);
  const bool narrow = g.N <= 64 || tiles128 < 384;
  const int tiles = narrow ? ((g.M + BM - 1) / BM) * ((g.N + 63) / 64) : tiles128;
  ifseg_prof_begin(IFSEG_K_CONV, (hipStream_t)stream, 2.0 * g.M * g.N * g.K, 2.0 * ((double)B * H * W * Cin + (double)g.N * g.K + (double)g.M * g.N));
  const bool two_stage = tiles <= two_stage_max();
  const dim3 grid(tiles, 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (narrow) LAUNCH2(A_CONV, false, 64);
  else LAUNCH2(A_CONV, false, 128);
  ifseg_prof_end(IFSEG_K_CONV, (hipStream_t)stream);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_abi_version(void) { return IFSEG_ABI_VERSION; }
// every object of the library reports its own measurement switches; any one of them marks the build
int ifseg_exp_attention();
extern "C" int ifseg_experimental_build(void) { return ifseg_exp_attention(); }
