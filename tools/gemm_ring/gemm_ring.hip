// Persistent bf16 MFMA GEMM for gfx950: one workgroup per CU walks its output tiles, operand k-tiles stream HBM -> LDS by
// LDS-DMA through a RING of S stages that runs ahead of the MFMAs ACROSS tile boundaries (the first stages of a workgroup's
// next tile are in flight under the epilogue of the current one), completion counted by hand (s_waitcnt vmcnt(n), never 0
// in steady state), one raw s_barrier per k-step.
//
// Why (DESIGN.md "Round 5"): the one-tile-per-workgroup kernel of gemm.hip pays, on the shapes of the model (M = 8480, K = 768:
// 12 k-steps per tile), one exposed L2 / HBM round trip per k-step (two stages: the request for step i+1 is issued when
// step i starts) and a fill + drain of 13-22 us per launch.  Here S-1 k-steps of operands are requested ahead, a workgroup
// owns all of a CU's LDS, and tile shape is a template parameter (larger tiles = fewer operand bytes per flop through the
// L2 -> LDS path, which is what bounds the k-loop).
//
// Layouts and epilogues are those of gemm.hip (reference ops: F.linear in unify_multihead_attention.py:327-346,513 and
// unify_transformer_layer.py:279-283,556-560; convolutions of resnet.py:117-137): NT / NN / TN / implicit-GEMM conv, bias,
// column-range alpha, residual, ReLU, fp32 / bf16 output, accumulate, row-dot, GELU + LayerNorm backward, grouped TN with
// fused column sums.  Same MFMA sequence per output element as gemm.hip => bit-identical results.
#include <cstdlib>
#include <utility>
#include "gemm_common.h"
#include "prof.h"
#include "gemm_ring.h"

namespace {

__device__ __forceinline__ v4i32 uniform4(v4i32 r) {
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
  r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  return r;
}
template <int N>
__device__ __forceinline__ void vm_wait() {
#ifndef RING_NOWAIT
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// every earlier LDS access of this wave has completed, then rendezvous (LDS-DMA requests stay in flight across it)
__device__ __forceinline__ void lds_barrier() {
#ifndef RING_NOBAR
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <int TM_, int TN_, int TK_, int NWM_, int NWN_, int S_>
struct RingCfg {
  static constexpr int TM = TM_, TN = TN_, TK = TK_, NWM = NWM_, NWN = NWN_, S = S_;
  static constexpr int NW = NWM * NWN, THREADS = NW * 64;
  static constexpr int WTM = TM / NWM, WTN = TN / NWN, TI = WTM / 32, TJ = WTN / 32;
  static constexpr int A_BYTES = TM * TK * 2, B_BYTES = TN * TK * 2, STAGE = A_BYTES + B_BYTES;
  static constexpr int LA = A_BYTES / 1024 / NW, LB = B_BYTES / 1024 / NW;   // 1-KiB DMA pieces per wave and stage
  static constexpr int P = LA + LB;
  static constexpr int KSTEPS = TK / 16;
  static constexpr int KS_SUB = TK * 256;                                    // bytes of one 128-column k-strided sub-tile
  static constexpr int RING = S * STAGE, LDS = RING + NW * 4096;         // ring + the epilogue's parking area
  static_assert(TJ == 2, "a wave owns 64 output columns (one head; the staged epilogue writes 128-byte row segments)");
  static_assert(LA * 1024 * NW == A_BYTES && LB * 1024 * NW == B_BYTES, "DMA pieces must divide evenly among the waves");
  static_assert((S - 2) * P <= 63 && S >= 2, "vmcnt is a 6-bit counter");
  static_assert(KSTEPS == 1 || KSTEPS % 2 == 0, "fragment sets alternate per k-slice");
  static_assert(LDS <= 160 * 1024, "LDS per CU");
};

// A workgroup's tiles: w = blockIdx.x + j * gridDim.x, remapped so that the workgroups of one XCD hold neighbouring tiles.
template <class C, int AMODE, bool B_KS, bool COLSUM, bool EPI_GLN>
__device__ __forceinline__ void ring_run(const GemmArgs* __restrict__ probs, const int* __restrict__ starts, const int nprob,
                                         const int total) {
  constexpr int TM = C::TM, TN = C::TN, TK = C::TK, S = C::S, TI = C::TI, LA = C::LA, LB = C::LB, P = C::P;
  constexpr int A_BYTES = C::A_BYTES, STAGE = C::STAGE, KSTEPS = C::KSTEPS;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[C::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::NWN, wn = wave % C::NWN;
  const int ntw = ((int)blockIdx.x < total) ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  auto tile_of = [&](int j, int& pid, int& m0, int& n0) {
    const int id = xcd_remap((int)blockIdx.x + j * (int)gridDim.x, total);
    pid = 0;
    for (int i = 1; i < nprob; ++i) pid = (id >= starts[i]) ? i : pid;
    const GemmArgs& g = probs[pid];
    const int tiles_n = (g.N + TN - 1) / TN, t = id - starts[pid];
    m0 = (t / tiles_n) * TM;
    n0 = (t % tiles_n) * TN;
  };

  // measurement only (-DRING_ABLATE=n builds, wrong results): leave out the MFMAs (1), the operand requests (2), the fragment reads (4)
#ifndef RING_ABLATE
#define RING_ABLATE 0
#endif
  constexpr int ablate = RING_ABLATE;
  // ------------------------------------------------------------------ loader (runs S-1 k-steps ahead of the MFMAs)
  int lj = 0, lkt = 0, lnk = 0, lK = 0, ls = 0;
  bool lvalid = ntw > 0;
  v4i32 rsA, rsB;
  unsigned offA[LA], offB[LB], kadvA = 0, kadvB = 0;
  int cv_base[AMODE == A_CONV ? LA : 1], cv_iy0[AMODE == A_CONV ? LA : 1], cv_ix0[AMODE == A_CONV ? LA : 1];
  int cky = 0, ckx = 0, cc0 = 0, cvW = 0, cvH = 0, cvC = 0, cvKW = 0;
  bool cv_interior = false;
  auto loader_tile = [&](int j) {
    int pid, m0, n0;
    tile_of(j, pid, m0, n0);
    const GemmArgs& g = probs[pid];
    rsA = make_rsrc(g.A, g.nrecA);
    rsB = make_rsrc(g.B, g.nrecB);
    lK = g.K;
    lnk = (g.K + TK - 1) / TK;
    kadvA = (AMODE == A_KS) ? (unsigned)TK * g.lda * 2 : TK * 2;
    kadvB = B_KS ? (unsigned)TK * g.ldb * 2 : TK * 2;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int p = wave * LA + i;
      if (AMODE == A_KS) {
        int kr, col;
        ks_src(p % (TK / 4), lane, kr, col);
        col += (p / (TK / 4)) * 128;
        offA[i] = (m0 + col < g.M) ? (unsigned)(((long long)kr * g.lda + m0 + col) * 2) : OOB;
      } else {
        int row, c;
        kct_src<TK>(p, lane, row, c);
        const int m = m0 + row;
        if (AMODE == A_KC) {
          offA[i] = (m < g.M) ? (unsigned)(((long long)m * g.lda + c * 8) * 2) : OOB;
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
      const int p = wave * LB + i;
      if (B_KS) {
        int kr, col;
        ks_src(p % (TK / 4), lane, kr, col);
        col += (p / (TK / 4)) * 128;
        offB[i] = (n0 + col < g.N) ? (unsigned)(((long long)kr * g.ldb + n0 + col) * 2) : OOB;
      } else {
        int row, c;
        kct_src<TK>(p, lane, row, c);
        const int n = n0 + row;
        offB[i] = (n < g.N) ? (unsigned)(((long long)n * g.ldb + c * 8) * 2) : OOB;
      }
    }
    if (AMODE == A_CONV) {
      cky = ckx = cc0 = 0;
      cvW = g.cW; cvH = g.cH; cvC = g.cC; cvKW = g.cKW;
      cv_interior = g.cPad == 0 && g.cKW == 1;
    }
  };
  // request piece group Q (of G = KSTEPS groups) of k-tile lkt of the loader's tile into stage ls; the last group steps the
  // loader.  A stage's pieces are spread over the k-slices of a k-step so that every MFMA group carries one or two of them.
  auto issue_group = [&](auto qtag) {
    constexpr int Q = decltype(qtag)::value, G = KSTEPS;
    const unsigned dA = lds0 + ls * STAGE + wave * (LA * 1024);
    const unsigned dB = lds0 + ls * STAGE + A_BYTES + wave * (LB * 1024);
    const int krem = lK - lkt * TK;                       // valid k in this tile (>= TK except on a tail)
    const bool tail = krem < TK;
    const int delta = (AMODE == A_CONV) ? ((cky * cvW + ckx) * cvC + cc0) * 2 : 0;   // wave-uniform offset of this tap / channel block
    const unsigned ka = (unsigned)lkt * kadvA, kb = (unsigned)lkt * kadvB;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if ((p * G) / P != Q || (ablate & 2)) continue;
      if (p < LA) {
        const int i = p;
        unsigned v;
        if (AMODE == A_CONV) {
          const bool ok = cv_interior ? cv_iy0[i] >= 0
                                      : ((unsigned)(cv_iy0[i] + cky) < (unsigned)cvH && (unsigned)(cv_ix0[i] + ckx) < (unsigned)cvW);
          v = ok ? (unsigned)(cv_base[i] + delta) : OOB;
        } else {
          v = offA[i] + ka;
          if (AMODE == A_KC && tail) {
            int row, c;
            kct_src<TK>(wave * LA + i, lane, row, c);
            if (c * 8 >= krem) v = OOB;
          }
        }
        lds_dma16(rsA, dA + i * 1024, v);
      } else {
        const int i = p - LA;
        unsigned v = offB[i] + kb;
        if (!B_KS && tail) {
          int row, c;
          kct_src<TK>(wave * LB + i, lane, row, c);
          if (c * 8 >= krem) v = OOB;
        }
        lds_dma16(rsB, dB + i * 1024, v);
      }
    }
    if (Q == G - 1) {
      if (AMODE == A_CONV) {
        cc0 += TK;
        if (cc0 >= cvC) { cc0 = 0; if (++ckx == cvKW) { ckx = 0; ++cky; } }
      }
      ls = (ls + 1 == S) ? 0 : ls + 1;
      if (++lkt == lnk) {
        lkt = 0;
        if (++lj < ntw) loader_tile(lj); else lvalid = false;
      }
    }
  };
  // The same for a FULL k-tile that is not the last of the loader's tile, with nothing to decide: the k offset rides in the
  // instruction's scalar offset (no per-piece VALU), no tail test, no tile switch.  (A k-tail must go through issue_group: its
  // zero padding comes from the per-lane offset.)
  auto issue_group_fast = [&](auto qtag) {
    constexpr int Q = decltype(qtag)::value, G = KSTEPS;
    const unsigned dA = lds0 + ls * STAGE + wave * (LA * 1024);
    const unsigned dB = dA - wave * (LA * 1024) + A_BYTES + wave * (LB * 1024);
    const unsigned ka = (unsigned)lkt * kadvA, kb = (unsigned)lkt * kadvB;
    const int delta = (AMODE == A_CONV) ? ((cky * cvW + ckx) * cvC + cc0) * 2 : 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if ((p * G) / P != Q || (ablate & 2)) continue;
      if (p < LA) {
        const int i = p;
        if (AMODE == A_CONV) {
          const bool ok = cv_interior ? cv_iy0[i] >= 0
                                      : ((unsigned)(cv_iy0[i] + cky) < (unsigned)cvH && (unsigned)(cv_ix0[i] + ckx) < (unsigned)cvW);
          lds_dma16(rsA, dA + i * 1024, ok ? (unsigned)(cv_base[i] + delta) : OOB);
        } else {
          lds_dma16_s(rsA, dA + i * 1024, offA[i], ka);
        }
      } else {
        const int i = p - LA;
        lds_dma16_s(rsB, dB + i * 1024, offB[i], kb);
      }
    }
    if (Q == G - 1) {
      if (AMODE == A_CONV) {
        cc0 += TK;
        if (cc0 >= cvC) { cc0 = 0; if (++ckx == cvKW) { ckx = 0; ++cky; } }
      }
      ls = (ls + 1 == S) ? 0 : ls + 1;
      ++lkt;
    }
  };
  auto issue_stage = [&]() {
    [&]<int... Q>(std::integer_sequence<int, Q...>) { (issue_group(std::integral_constant<int, Q>{}), ...); }
    (std::make_integer_sequence<int, KSTEPS>{});
  };

  // ------------------------------------------------------------------ consumer
  // Register double-buffered fragments: the reads of k-slice s+1 are requested before the MFMAs of slice s; the barrier of a
  // k-step sits BEFORE its last slice, so the first fragments of the next stage are requested under the last MFMAs of this one.
  f32x16 acc[TI][2];
  f32x16 accb[COLSUM ? TI : 1];
  bf16x8 fa[2][TI], fb[2][2];
  auto ldfrag = [&](auto settag, int st, int ks) {
    constexpr int SET = decltype(settag)::value;
    if (ablate & 4) return;
    const unsigned char* sA = smem + st * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int r = wm * C::WTM + i * 32;
      fa[SET][i] = (AMODE == A_KS) ? frag_ks(sA + (r >> 7) * C::KS_SUB, r & 127, ks, lane) : frag_kct<TK>(sA, r, ks, lane);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wn * 64 + j * 32;
      fb[SET][j] = B_KS ? frag_ks(sB + (r >> 7) * C::KS_SUB, r & 127, ks, lane) : frag_kct<TK>(sB, r, ks, lane);
    }
  };
  auto wait_younger = [&](int younger) {
    if (S >= 6 && younger >= 4) vm_wait<(S >= 6 ? 4 : 0) * P>();
    else if (S >= 5 && younger >= 3) vm_wait<(S >= 5 ? 3 : 0) * P>();
    else if (S >= 4 && younger >= 2) vm_wait<(S >= 4 ? 2 : 0) * P>();
    else if (S >= 3 && younger >= 1) vm_wait<(S >= 3 ? 1 : 0) * P>();
    else vm_wait<0>();
  };
  int issued = 0, it = 0, cs = 0;       // stages requested in full; global k-step index; stage being consumed
  bool lopen = false;                   // a stage is partly requested (its group 0 went out after the last barrier)
  if (lvalid) loader_tile(0);
#pragma unroll 1
  for (int p = 0; p < S; ++p)
    if (lvalid) { issue_stage(); ++issued; }
  if (ntw > 0) {
    wait_younger(issued - 1);
    lds_barrier();
    ldfrag(std::integral_constant<int, 0>{}, 0, 0);
  }

#pragma unroll 1
  for (int cj = 0; cj < ntw; ++cj) {
    int pid, m0, n0;
    tile_of(cj, pid, m0, n0);
    const GemmArgs& g = probs[pid];
    const int nk = (g.K + TK - 1) / TK;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // COLSUM (weight-gradient GEMMs): the column sums of A (= the bias gradient) come out of the same A fragments through
    // one more MFMA against an all-ones B fragment, in the workgroups of the first column tile only
    const bool do_colsum = COLSUM && n0 == 0 && wn == 0 && (g.flags & IFSEG_GEMM_COLSUM);
    if (COLSUM) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
    }
    // the lane's bias values (4 consecutive columns per register group) are requested under the last k-step
    constexpr bool PRE_BIAS = AMODE != A_KS && !B_KS;
    uint2 biasr[PRE_BIAS ? 2 : 1][4];

    auto prefetch_bias = [&]() {
      if (PRE_BIAS && g.bias) {
        // hidden from the compiler's wait counting (a load it knows of makes it drain the whole ring at the first use);
        // columns beyond N read zeros through the descriptor; waited for by hand in front of the epilogue
        const v4i32 rsb = make_rsrc(g.bias, (unsigned)g.N * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const unsigned nb = (unsigned)(n0 + wn * 64 + j * 32 + 8 * rg + 4 * (lane >> 5)) * 2;
            asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(biasr[j][rg]) : "v"(nb), "s"(rsb) : "memory");
          }
      }
    };
    auto mma = [&](auto curtag) {
      constexpr int cur = decltype(curtag)::value;
      __builtin_amdgcn_sched_barrier(0);
      if (ablate & 1) {
#pragma unroll
        for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(fa[cur][i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[cur][j]));
      } else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
      }
      if constexpr (COLSUM) {
        if (do_colsum) {
          U128 one;
          one.w[0] = one.w[1] = one.w[2] = one.w[3] = 0x3F803F80u;      // eight bf16 1.0
#pragma unroll
          for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one.b, fa[cur][i], accb[i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    int kt = 0;
#pragma unroll 1
    while (kt < nk) {
      // (the descriptors ARE wave-uniform; restated for the compiler, whose analysis loses it across this loop's back edges --
      // the inline-asm "s" operands would not compile -- folds to nothing)
      rsA = uniform4(rsA); rsB = uniform4(rsB);
      // hot k-step: the ring is full (a stage is open: its group 0 went out after the last barrier), the open k-tile and the
      // next one are FULL k-tiles of the loader's current tile, and another k-step follows.  Its requests take the scalar
      // k offset (no per-piece VALU, no tail test, no tile switch) and its wait is the steady-state constant; everything
      // else goes the general way (loader at a tile boundary or on a k-tail, ring filling or draining, last step).
      const bool more = !(cj == ntw - 1 && kt == nk - 1);     // another k-step follows (maybe of the next tile)
      const bool hot = lopen && lvalid && more && lkt + 2 <= lK / TK;
      const int ns = (cs + 1 == S) ? 0 : cs + 1;
      if (kt == nk - 1) prefetch_bias();
      [&]<int... KS>(std::integer_sequence<int, KS...>) {
        ([&] {
          constexpr int ks = KS, cur = KS & 1, nxt = cur ^ 1;
          if constexpr (ks < KSTEPS - 1) {
            ldfrag(std::integral_constant<int, nxt>{}, cs, ks + 1);
            if (hot) {
              issue_group_fast(std::integral_constant<int, ks + 1>{});
              if (ks + 1 == KSTEPS - 1) ++issued;
            } else if (lopen) {
              issue_group(std::integral_constant<int, ks + 1>{});
              if (ks + 1 == KSTEPS - 1) { lopen = false; ++issued; }
            }
          } else {
            // every wave's pieces of the next stage have landed, and every wave holds its last fragments of this one:
            // this stage is free for the request after next
            if (hot) vm_wait<(S - 2) * P>();
            else if (more) wait_younger(issued - (it + 2));
            lds_barrier();
            if (more) ldfrag(std::integral_constant<int, nxt>{}, ns, 0);
            if (hot) {
              issue_group_fast(std::integral_constant<int, 0>{});
              if (KSTEPS == 1) ++issued;
            } else if (lvalid) {
              issue_group(std::integral_constant<int, 0>{});
              if (KSTEPS == 1) ++issued; else lopen = true;
            }
          }
          mma(std::integral_constant<int, cur>{});
        }(), ...);
      }(std::make_integer_sequence<int, KSTEPS>{});
      cs = ns;
      ++kt; ++it;
    }

    // ---------------------------------------------------------------- epilogue of tile cj
    if constexpr (COLSUM) {
      // every row of accb holds sum_k A[k][m]; lanes 0..31 carry row 0 in register 0: db (bf16) sits right behind
      // dW [M x ldc] in the gradient arena
      if (do_colsum && lane < 32) {
        bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + (long long)g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int m = m0 + wm * C::WTM + i * 32 + lane;
          if (m < g.M) cb[m] = f2bf(accb[i][0] + ((g.flags & IFSEG_GEMM_ACCUMULATE) ? bf2f(cb[m]) : 0.f));
        }
      }
    }
    if (PRE_BIAS && g.bias) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(biasr[0][0]), "+v"(biasr[0][1]), "+v"(biasr[0][2]), "+v"(biasr[0][3]), "+v"(biasr[PRE_BIAS ? 1 : 0][0]),
                     "+v"(biasr[PRE_BIAS ? 1 : 0][1]), "+v"(biasr[PRE_BIAS ? 1 : 0][2]), "+v"(biasr[PRE_BIAS ? 1 : 0][3])
                   :: "memory");
    }
    const bool relu = g.flags & IFSEG_GEMM_RELU, out_f32 = g.flags & IFSEG_GEMM_OUT_F32, accum = g.flags & IFSEG_GEMM_ACCUMULATE;
    const bf16_t* Rb = g.resid;
    // bf16 output leaves through LDS (see gemm.hip): a lane owns a ROW of the MFMA tile, so direct stores write 32-byte
    // pieces of 32 rows per instruction; each wave parks 32 x 64 outputs in its own 4 KiB of the stage consumed last
    // (16-byte chunks XOR-swizzled by the row) and writes them back as 8 full 128-byte row segments per instruction.
    // The parking area lies behind the ring and is private to the wave: the epilogue needs no barrier and the ring keeps
    // filling under it.
    const bool lds_out = !out_f32 && !accum && !(g.ldc & 7) && !((size_t)g.C & 15);
    unsigned char* sOut = smem + C::RING + wave * 4096;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int mrow0 = m0 + wm * C::WTM + i * 32;
      const int m = mrow0 + (lane & 31);
      const bool mvalid = m < g.M;
      float dsum = 0.f;
      float gl_mu = 0.f, gl_rs = 0.f, gl_c1 = 0.f, gl_c2 = 0.f;
      if constexpr (EPI_GLN) {
        if (mvalid) { gl_mu = g.gln_mean[m]; gl_rs = g.gln_rstd[m]; gl_c1 = g.gln_c[2 * m]; gl_c2 = g.gln_c[2 * m + 1]; }
      }
      if (lds_out || mvalid) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nj = n0 + wn * 64 + j * 32;
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
              const uint2 rw = *reinterpret_cast<const uint2*>(Rb + (long long)m * g.ldr + n);
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
              float* cp = reinterpret_cast<float*>(g.C) + (long long)m * g.ldc + n;
              float4 o = make_float4(v[0], v[1], v[2], v[3]);
              if (accum) {
                const float4 p = *reinterpret_cast<float4*>(cp);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
              }
              *reinterpret_cast<float4*>(cp) = o;
            } else {
              bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long long)m * g.ldc + n;
              if (accum) {
                const uint2 pw = *reinterpret_cast<const uint2*>(cp);
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
            const int R = lane & 31;
#pragma unroll
            for (int rgp = 0; rgp < 2; ++rgp) {
              const auto p0 = __builtin_amdgcn_permlane32_swap(held[2 * rgp].x, held[2 * rgp + 1].x, false, false);
              const auto p1 = __builtin_amdgcn_permlane32_swap(held[2 * rgp].y, held[2 * rgp + 1].y, false, false);
              const int c = j * 4 + rgp * 2 + (lane >> 5);
              *reinterpret_cast<uint4*>(sOut + R * 128 + ((c ^ (R & 7)) << 4)) = make_uint4(p0[0], p1[0], p0[1], p1[1]);
            }
          }
        }
      }
      if (g.dot && mvalid) {
        // the wave's 64 columns are one head; lanes l and l + 32 hold the two interleaved halves of row m
        dsum += __shfl_xor(dsum, 32);
        const int hd = (n0 + wn * 64) >> 6;
        if (lane < 32 && (hd << 6) < g.N)
          g.dot_out[((long long)(m / g.dot_T) * (g.N >> 6) + hd) * g.dot_T + (m % g.dot_T)] = dsum;
      }
      if (lds_out) {
        const int c = lane & 7, nc = n0 + wn * 64 + c * 8;
        bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + nc;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int R = it * 8 + (lane >> 3), mm = mrow0 + R;
          const uint4 v = *reinterpret_cast<const uint4*>(sOut + R * 128 + ((c ^ (R & 7)) << 4));
          if (mm < g.M && nc + 8 <= g.N) *reinterpret_cast<uint4*>(cb + (long long)mm * g.ldc) = v;
        }
      }
    }
  }
}

struct RingArgs1 { int start[2]; GemmArgs p; };

template <class C, int AMODE, bool B_KS, bool EPI_GLN>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_ring_kernel(RingArgs1 a, int total) {
  ring_run<C, AMODE, B_KS, false, EPI_GLN>(&a.p, a.start, 1, total);
}
template <class C>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_ring_group_kernel(GroupArgs ga) {
  ring_run<C, A_KS, true, true, false>(ga.p, ga.start, ga.n, ga.total);
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// tile configurations (id = what IFSEG_GEMM_RING / the selection heuristic names)
// (LDS = ring + 4 KiB of parking area per wave)
using C1 = RingCfg<128, 128, 64, 2, 2, 3>;    // 112 KiB
using C3 = RingCfg<256, 128, 64, 4, 2, 2>;    // 128 KiB, 8 waves of 64 x 64
using C4 = RingCfg<128, 256, 64, 2, 4, 2>;    // 128 KiB, 8 waves of 64 x 64
using C5 = RingCfg<256, 256, 32, 2, 4, 4>;    // 160 KiB, 8 waves of 128 x 64, 32-deep k-steps
using C6 = RingCfg<128, 128, 64, 2, 2, 2>;    //  80 KiB: two workgroups per CU
using C7 = RingCfg<256, 256, 64, 2, 4, 2>;    // 160 KiB, 8 waves of 128 x 64
using C8 = RingCfg<256, 128, 32, 4, 2, 2>;    //  80 KiB, 8 waves of 64 x 64, 32-deep k-steps (round 6: VERDICT r5 item 1 b)
using C9 = RingCfg<256, 128, 32, 4, 2, 3>;    // 104 KiB
using C10 = RingCfg<128, 256, 32, 2, 4, 2>;   //  80 KiB

template <class C, int AMODE, bool B_KS, bool EPI_GLN>
int launch1(const GemmArgs& g, int wgs_per_cu, hipStream_t s) {
  RingArgs1 a{};
  a.p = g;
  const int total = ((g.M + C::TM - 1) / C::TM) * ((g.N + C::TN - 1) / C::TN);
  a.start[0] = 0; a.start[1] = total;
  int grid = wgs_per_cu > 0 ? num_cus() * wgs_per_cu : total;       // 0: one workgroup per tile (not persistent)
  if (grid > total) grid = total;
  hipLaunchKernelGGL((gemm_ring_kernel<C, AMODE, B_KS, EPI_GLN>), dim3(grid), dim3(C::THREADS), 0, s, a, total);
  return 0;
}

template <int AMODE, bool B_KS, bool EPI_GLN>
int launch_cfg(int cfg, const GemmArgs& g, hipStream_t s) {
  switch (cfg) {
    case 1: return launch1<C1, AMODE, B_KS, EPI_GLN>(g, 1, s);
    case 3: return launch1<C3, AMODE, B_KS, EPI_GLN>(g, 1, s);
    case 4: return launch1<C4, AMODE, B_KS, EPI_GLN>(g, 1, s);
    case 5: return launch1<C5, AMODE, B_KS, EPI_GLN>(g, 1, s);
    case 6: return launch1<C6, AMODE, B_KS, EPI_GLN>(g, 2, s);
    case 7: return launch1<C7, AMODE, B_KS, EPI_GLN>(g, 1, s);
  }
  return IFSEG_ERR_BAD_ARG;
}

template <class C>
int launch_group(GroupArgs ga, int max_workgroups, hipStream_t s) {
  int total = 0;
  for (int i = 0; i < ga.n; ++i) {
    ga.start[i] = total;
    total += ((ga.p[i].M + C::TM - 1) / C::TM) * ((ga.p[i].N + C::TN - 1) / C::TN);
  }
  ga.start[ga.n] = total;
  ga.total = total;
  // one tile per workgroup and CU, or not at all: a second, nearly empty round of 133-k-step tiles costs more than the wider
  // tile saves (a decoder layer's seven products are 288 tiles of 256 x 128) -- the caller falls back to the tile kernel
  if (total > num_cus() && !getenv("IFSEG_GEMM_RING_GROUP_ANY")) return 1;
  int grid = num_cus();
  if (max_workgroups > 0 && max_workgroups < grid) grid = max_workgroups >= 8 ? (max_workgroups & ~7) : max_workgroups;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(gemm_ring_group_kernel<C>, dim3(grid), dim3(C::THREADS), 0, s, ga);
  return 0;
}

}  // namespace

int gemm_ring_group_launch(const void* group_args, int cfg, int max_workgroups, void* stream) {
  const GroupArgs& ga = *reinterpret_cast<const GroupArgs*>(group_args);
  hipStream_t s = (hipStream_t)stream;
  switch (cfg) {
    case 1: return launch_group<C1>(ga, max_workgroups, s);
    case 3: return launch_group<C3>(ga, max_workgroups, s);
    case 4: return launch_group<C4>(ga, max_workgroups, s);
    case 8: return launch_group<C8>(ga, max_workgroups, s);
    case 9: return launch_group<C9>(ga, max_workgroups, s);
    case 10: return launch_group<C10>(ga, max_workgroups, s);
  }
  return IFSEG_ERR_BAD_ARG;
}

int gemm_ring_launch(const void* gemm_args, int amode, int b_ks, int epi_gln, int cfg, void* stream) {
  const GemmArgs& g = *reinterpret_cast<const GemmArgs*>(gemm_args);
  hipStream_t s = (hipStream_t)stream;
  if (epi_gln) return launch_cfg<A_KC, true, true>(cfg, g, s);
  if (amode == A_KC && !b_ks) return launch_cfg<A_KC, false, false>(cfg, g, s);
  if (amode == A_KC && b_ks) return launch_cfg<A_KC, true, false>(cfg, g, s);
  if (amode == A_CONV && !b_ks) return launch_cfg<A_CONV, false, false>(cfg, g, s);
  if (amode == A_KS && b_ks) return launch_cfg<A_KS, true, false>(cfg, g, s);
  return IFSEG_ERR_BAD_ARG;
}


int ifseg_exp_gemm_ring() {
#if RING_ABLATE != 0 || defined(RING_NOBAR) || defined(RING_NOWAIT)
  return 1;
#else
  return 0;
#endif
}
