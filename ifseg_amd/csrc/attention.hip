// Fused position-biased attention for SegOFA on gfx950 (forward + backward).
//
// Reference semantics (unify_multihead_attention.py:346,459-501 and the bias
// construction in encoder_module.py:757-809 / decoder_module.py:335-366,601-631):
//     S = (q*scaling) k^T + abs_pos_bias + rel_pos_bias (+ causal -inf mask)
//     P = softmax_fp32(S);  O = P v
// where abs_pos_bias[h] = (pos_q*pos_scaling) pos_k^T is batch invariant and
// rel_pos_bias[h,i,j] = table[bucket(i,j), h].
//
// MI355X formulation: the abs-pos term is folded into the contraction,
// S = [q | pos_q] . [k | pos_k]^T (head dim 64 -> 128), so neither the [H,T,S]
// bias nor the [B*H,T,S] scores ever exist in HBM, and its gradient falls out of
// the same dQ/dK MFMAs.  The rel-pos term is a per-head delta table held in LDS
// and indexed arithmetically (image grid: code_i - code_j; text: i - j).
// Everything is computed "swapped" (S^T = K Q^T, O^T = V^T P^T) so that a query
// row lives in one lane: softmax statistics are per-lane scalars, P feeds the
// second MFMA straight from registers, and V / K are consumed through
// ds_read_b64_tr_b16 with a free choice of key order.
//
// Token order inside a sequence: [grid tokens 0..P-1 | tail tokens P..T-1].
// Encoder: grid = image patches, tail = text.  Decoder: grid = patches, tail =
// the single bos token (the host permutes bos to the end; `causal` uses
// "tail-first" original order: a tail key is visible to every grid query).
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {

struct AttnArgs {
  const bf16_t *q, *k, *v, *pq, *pk;
  bf16_t* o;
  float* lse;
  int B, H, T, S;
  long long q_bs, k_bs, v_bs, o_bs;
  int ldq, ldk, ldv, ldo, ldpq, ldpk;
  int rel_mode, P, code_bias, n2d, Lt, causal;
  const int* gcode;
  const float *rel2d, *rel1d, *relx, *dense;
  int dense_ld;               // row stride of `dense` [H, T, dense_ld] (>= S; a multiple of 4 enables the 16-byte seed loads)
  // backward
  const bf16_t* dO; long long do_bs; int lddo;
  const float* delta;
  bf16_t *dq, *dk, *dv; long long dq_bs, dk_bs, dv_bs; int lddq, lddk, lddv;
  bf16_t *dpq, *dpk;         // [B, T, H*64] / [B, S, H*64] bf16 per-batch partials
  float *drel2d_part, *drel1d_part, *drelx_part;  // [H][nparts][n]
  int nparts;
  const float* gain;          // [H] per-head output gain c_attn (fp32: the optimizer's master copy; may be null)
  float* dgain_rows;          // optional [B,H,T]: sum_j P_ij dP_ij = dO_i . (P V)_i, the per-row terms of d c_attn (no division by c_attn)
  float dq_scale, dpq_scale;
  int grid_w;                 // width of the token grid (row-aligned diagonal reduction when 32)
};

constexpr int KT_BYTES = 64 * 256;  // K_ext tile  [64 keys][128] bf16
constexpr int VT_BYTES = 64 * 128;  // V tile      [64 keys][64]  bf16
constexpr int DKV_STAGE_BYTES = 32 * 256 + 32 * 128 + 256;  // dK/dV kernel: Q_ext, dO, lse|delta of 32 queries
static_assert(2 * DKV_STAGE_BYTES == KT_BYTES + VT_BYTES + 512, "host-side LDS size formula");
constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
// forward: the running reference maximum is only raised when a tile exceeds it by more than this
// (log2 units), so the O / l rescale is a rare wave-uniform branch; P stays <= 2^8 in between
constexpr float LAZY_MAX_SLACK = 8.f;

// LDS images that are read BOTH row-wise (ds_read_b128: 16 rows x one 16-byte chunk per lane
// group) and transposed (ds_read_b64_tr_b16: 4 consecutive rows x one 64-byte granule):
// the chunk index is XOR-ed with a bit-rotated row id so that 16 consecutive rows hit 16
// different 16-byte slots of the 256-byte bank line AND 4 consecutive rows hit 4 different
// 64-byte bank quarters.
// K_ext / Q_ext tile: 256-byte rows (16 chunks)
__device__ __forceinline__ int kx_off(int r, int c) {
  const int f = ((r & 3) << 2) | ((r >> 2) & 3);
  return r * 256 + ((c ^ f) << 4);
}
// V / dO tile: 128-byte rows (8 chunks, two rows per bank line)
__device__ __forceinline__ int vx_off(int r, int colbyte) {
  const int u = r >> 1, hsw = ((u & 1) << 2) | ((u >> 1) & 3);
  return r * 128 + ((((colbyte >> 4) ^ hsw) & 7) << 4) + (colbyte & 15);
}

// Marks a register fragment as used here.  Fragments loaded from global memory in a kernel prologue and first read
// inside the main loop leave their loads "pending" at the loop header in the compiler's wait-count model; it then puts
// s_waitcnt vmcnt(N) with small N in front of their uses INSIDE the loop, and from the second iteration on those waits
// hit the loop's own prefetch (global loads / LDS-DMA for the next tile), serialising it with the MFMAs.
__device__ __forceinline__ void consume_frag(const bf16x8& f) {
  typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
  asm volatile("" :: "v"(__builtin_bit_cast(u32x4_t, f)));
}

struct RelCtx {
  const float* tbl;   // LDS copy of rel2d[h]
  const int* gc;      // LDS copy of gcode
  const float* rel1d; // global, rel1d[h]
  float relx0, relx1;
};

// ---------------------------------------------------------------- forward

// Global -> LDS copy of a small table by a 256-thread workgroup with eight loads in flight per thread (a plain
// `for (i = tid; i < n; i += 256)` loop issues one load per round trip: 16 serial L2 latencies for a 63 x 63 table).
template <typename T, bool REVERSE = false, int NT = 256>
__device__ __forceinline__ void stage_table(T* dst, const T* __restrict__ src, int n, int tid) {
  for (int i0 = 0; i0 < n; i0 += 8 * NT) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + tid; v[u] = i < n ? src[i] : T(0); }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + tid; if (i < n) dst[REVERSE ? n - 1 - i : i] = v[u]; }
  }
}


// Epilogue of a 32 x 32 accumulator tile whose row (query / key) is the lane: element r of lane (half, x) is column
// (r&3) + 8*(r>>2) + 4*half, i.e. 8-byte runs.  The two lanes of a row exchange one run each (v_permlane32_swap), so
// that every lane stores 16 contiguous bytes: half as many store instructions (the store tail is issue-bound).
__device__ __forceinline__ void store_tile_bf16(bf16_t* rowp, const f32x16& acc, float scale, int half, bool valid) {
#pragma unroll
  for (int rgp = 0; rgp < 2; ++rgp) {
    const int e0 = rgp * 8, e1 = rgp * 8 + 4;
    const unsigned x0 = pack2bf(acc[e0] * scale, acc[e0 + 1] * scale), x1 = pack2bf(acc[e0 + 2] * scale, acc[e0 + 3] * scale);
    const unsigned y0 = pack2bf(acc[e1] * scale, acc[e1 + 1] * scale), y1 = pack2bf(acc[e1 + 2] * scale, acc[e1 + 3] * scale);
    const auto p0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
    const auto p1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    // lanes 0..31: columns 16*rgp .. +7 (own run, partner's run); lanes 32..63: columns 16*rgp + 8 .. +15
    if (valid) *reinterpret_cast<uint4*>(rowp + 16 * rgp + 8 * half) = make_uint4(p0[0], p1[0], p0[1], p1[1]);
  }
}

template <bool HAS_POS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto sKb = [&](int buf) { return smem + buf * (KT_BYTES + VT_BYTES); };
  auto sVb = [&](int buf) { return smem + buf * (KT_BYTES + VT_BYTES) + KT_BYTES; };
  float* sTbl = reinterpret_cast<float*>(smem + 2 * (KT_BYTES + VT_BYTES));
  int* sGc = reinterpret_cast<int*>(sTbl + ((a.n2d + 3) & ~3));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int nq = (a.T + 127) >> 7;
  int qt, h, b;
  if (a.causal) {     // later query tiles see more keys: longest workgroups first
    int rank, bh;
    causal_order(blockIdx.x, a.H * a.B, &rank, &bh);
    qt = nq - 1 - rank; h = bh % a.H; b = bh / a.H;
  } else {
    const int bid = xcd_remap(blockIdx.x, nq * a.H * a.B);
    qt = bid % nq; h = (bid / nq) % a.H; b = bid / (nq * a.H);
  }
  const int q0 = qt * 128, qw = q0 + wave * 32;
  const int qi = qw + (lane & 31);           // this lane's query row
  const bool qvalid = qi < a.T;
  const int qrow = qvalid ? qi : a.T - 1;
  constexpr int NKS = HAS_POS ? 8 : 4;

  // ---- Q_ext fragments (B operand of S^T = K Q^T): lane holds Q[qi][ks*16 + half*8 .. +8]
  bf16x8 qf[NKS];
  {
    const bf16_t* qp = a.q + (long long)b * a.q_bs + (long long)qrow * a.ldq + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(qp + ks * 16); qf[ks] = u.b; }
    if (HAS_POS) {
      const bf16_t* pp = a.pq + (long long)qrow * a.ldpq + h * 64 + half * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(pp + ks * 16); qf[4 + ks] = u.b; }
    }
  }
  // ---- rel-pos tables into LDS
  if (a.rel_mode) {
    // the table is kept REVERSED in LDS (sTbl[n2d - 1 - i] = rel2d[i]): a lane is a query and its registers are keys of
    // increasing x, i.e. of decreasing table index; reversed, the 16 seeds of a block are read in register order
    // (ds_read2_b32 pairs land in the accumulator tuple; in table order every seed needed a v_mov: ~60 per tile)
    stage_table<float, true>(sTbl, a.rel2d + (long long)h * a.n2d, a.n2d, tid);
    stage_table(sGc, a.gcode, a.P, tid);
  }
  const bool q_grid = a.rel_mode && qi < a.P;
  const int ciR = q_grid ? a.n2d - 1 - (a.gcode[qi] + a.code_bias) : 0;    // reversed-table index = ciR + cj
  const int ti = qi - a.P;
  const float relx0 = a.rel_mode ? a.relx[h * 2 + 0] : 0.f, relx1 = a.rel_mode ? a.relx[h * 2 + 1] : 0.f;
  const float* rel1d = a.rel_mode ? a.rel1d + (long long)h * (2 * a.Lt - 1) + (a.Lt - 1) : nullptr;
  const bool row32 = a.rel_mode && a.grid_w == 32;   // raster grid, 32 wide: a 32-key block is one grid row
  // any grid width that is a multiple of 8 (>= 32): a 32-key block spans at most two grid rows and every aligned group of
  // 8 keys -- the 4-key runs a lane holds -- lies inside one row: the seeds sit at constant offsets from ONE ADDRESS PER
  // GROUP instead of one per block (BASELINE configs[3]: 640 x 640 -> a 40-wide grid)
  const bool rowseg = a.rel_mode && !row32 && a.grid_w >= 32 && (a.grid_w & 7) == 0;

  // ---- tile schedule (causal: skip grid tiles wholly above the diagonal)
  const int ntile = (a.S + 63) >> 6;
  const int Pk = a.rel_mode || a.causal ? a.P : a.S;        // keys < Pk are "grid" keys
  const int ngrid = Pk >> 6;                                 // P % 64 == 0 enforced by the host
  int g_end = ngrid;
  if (a.causal) {
    int imax = (q0 < a.P) ? min(q0 + 127, a.P - 1) : -1;
    g_end = imax >= 0 ? min(ngrid, (imax >> 6) + 1) : 0;
  }
  const int nsched = g_end + (ntile - ngrid);
  auto tile_of = [&](int it) { return it < g_end ? it : ngrid + (it - g_end); };

  // ---- staging registers
  uint4 rk[4], rv[2];
  const bf16_t* kb_ = a.k + (long long)b * a.k_bs + h * 64;
  const bf16_t* vb_ = a.v + (long long)b * a.v_bs + h * 64;
  const bf16_t* pkb_ = HAS_POS ? a.pk + h * 64 : nullptr;
  auto load_kv = [&](int j0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7, j = j0 + r;
      rk[i] = make_uint4(0, 0, 0, 0); rk[2 + i] = make_uint4(0, 0, 0, 0); rv[i] = make_uint4(0, 0, 0, 0);
      if (j < a.S) {
        // wave-uniform base + 32-bit lane offset (saddr addressing): no per-lane 64-bit pointers kept across the loop
        rk[i] = *reinterpret_cast<const uint4*>(kb_ + (unsigned)(j * a.ldk + c * 8));
        if (HAS_POS) rk[2 + i] = *reinterpret_cast<const uint4*>(pkb_ + (unsigned)(j * a.ldpk + c * 8));
        rv[i] = *reinterpret_cast<const uint4*>(vb_ + (unsigned)(j * a.ldv + c * 8));
      }
    }
  };
  auto store_kv = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7;
      *reinterpret_cast<uint4*>(sKb(buf) + kx_off(r, c)) = rk[i];
      if (HAS_POS) *reinterpret_cast<uint4*>(sKb(buf) + kx_off(r, 8 + c)) = rk[2 + i];
      *reinterpret_cast<uint4*>(sVb(buf) + vx_off(r, c * 16)) = rv[i];
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
  float m_run = NEG_INF, l_run = 0.f;

  if (nsched > 0) { load_kv(tile_of(0) * 64); store_kv(0); }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) consume_frag(qf[ks]);
  __syncthreads();

  for (int it = 0; it < nsched; ++it) {
    const int cur = it & 1;
    const int j0 = tile_of(it) * 64;
    if (it + 1 < nsched) load_kv(tile_of(it + 1) * 64);
    const bool tile_grid = j0 < Pk;
    // wave-level causal skip: every key of a grid tile is beyond every query of this wave
    // (waves whose 32 queries all lie past T only help staging the tiles)
    const bool skip = (qw >= a.T) || (a.causal && tile_grid && (j0 > qw + 31 || qw >= a.P));
    if (!skip) {
      const int qw_u = __builtin_amdgcn_readfirstlane(qw);
      const bool wave_grid = qw_u + 31 < a.P;
      const bool gg = a.rel_mode && tile_grid && wave_grid && !a.dense;   // grid x grid: the bulk of self-attention
      // grid queries x text keys / text queries x grid keys: the bias is one scalar per head; no relative bias at all
      // (cross attention): both take the straight-line path below.  P % 32 == 0: a wave lies on one side.
      const bool cbias = a.rel_mode && !a.dense && ((wave_grid && !tile_grid) || (qw_u >= a.P && tile_grid));
      const bool plain = cbias || (!a.rel_mode && !a.causal && !a.dense);
      // a dense fp32 bias alone (resized-grid evaluation: the causal mask travels inside it): the lane's query row is fixed
      // and its 16 keys are four runs of four, so the accumulator is seeded with four 16-byte loads per 32-key block
      // (with `causal` the mask travels inside the dense bias as -inf; the flag then only drives the tile schedule)
      const bool dfast = a.dense && !a.rel_mode && !(a.dense_ld & 3);
      f32x16 s[2];
      auto s_mfma = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 kf = lds_read_b128(sKb(cur) + kx_off(kb * 32 + (lane & 31), ks * 2 + half));
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
          }
        }
      };
      if (gg && row32) {
        // the block's keys are one grid row (codes cjb + x): the lane's 16 bias values sit at constant
        // offsets from one table address and are loaded straight into the accumulator, so the MFMA adds them
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int cjb = sGc[j0 + kb * 32];
          const float* tp = sTbl + (ciR + cjb + 4 * half);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kb][rg * 4 + e] = tp[8 * rg + e];
        }
        s_mfma();
      } else if (gg && rowseg) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float* tp = sTbl + (ciR + sGc[j0 + kb * 32 + 8 * rg] + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kb][rg * 4 + e] = tp[e];
          }
        }
        s_mfma();
      } else if (dfast) {
        const float* drow = a.dense + ((long long)h * a.T + qrow) * a.dense_ld;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int jb = j0 + kb * 32 + 8 * rg + 4 * half;
            const float4 d4 = jb < a.dense_ld ? *reinterpret_cast<const float4*>(drow + jb) : make_float4(0.f, 0.f, 0.f, 0.f);
            s[kb][rg * 4 + 0] = d4.x; s[kb][rg * 4 + 1] = d4.y; s[kb][rg * 4 + 2] = d4.z; s[kb][rg * 4 + 3] = d4.w;
          }
        }
        s_mfma();
      } else {
        // (a separate MFMA chain: zero accumulators cost nothing, the first MFMA takes the constant; a chain shared
        // with the seeded path makes the compiler splat 32 registers on every tile before it knows the path)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        s_mfma();
        if (cbias) {
          const float c0 = tile_grid ? relx1 : relx0;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] += c0;
        }
      }
      // ---- bias + mask ; lane element (kb, r) <-> key j0 + kb*32 + (r&3) + 8*(r>>2) + 4*half
      // (scores stay in natural units; log2 e is folded into the exponent's fma)
      float mx = NEG_INF;
      if (gg) {
        if (!row32 && !rowseg) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int jb = j0 + kb * 32 + 8 * rg + 4 * half;
              const int4 cj = *reinterpret_cast<const int4*>(sGc + jb);
              s[kb][rg * 4 + 0] += sTbl[ciR + cj.x];
              s[kb][rg * 4 + 1] += sTbl[ciR + cj.y];
              s[kb][rg * 4 + 2] += sTbl[ciR + cj.z];
              s[kb][rg * 4 + 3] += sTbl[ciR + cj.w];
            }
          }
        }
        if (a.causal && j0 + 63 > qw_u) {      // only tiles crossing the diagonal hold masked elements
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int dj = j0 + kb * 32 + 8 * rg + 4 * half - qi;
#pragma unroll
              for (int e = 0; e < 4; ++e) s[kb][rg * 4 + e] = (dj + e > 0) ? NEG_INF : s[kb][rg * 4 + e];
            }
        }
        float m4[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) m4[e & 3] = fmaxf(m4[e & 3], s[kb][e]);
        mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      } else if (plain || dfast) {
        if (j0 + 64 > a.S) {           // last tile: keys past S are masked
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int jl = a.S - (j0 + kb * 32 + 8 * rg + 4 * half);     // element e is a real key iff e < jl
#pragma unroll
              for (int e = 0; e < 4; ++e) s[kb][rg * 4 + e] = (e < jl) ? s[kb][rg * 4 + e] : NEG_INF;
            }
        }
        float m4[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) m4[e & 3] = fmaxf(m4[e & 3], s[kb][e]);
        mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int jb = j0 + kb * 32 + 8 * rg + 4 * half;
            int4 cj = make_int4(0, 0, 0, 0);
            if (a.rel_mode && tile_grid) cj = *reinterpret_cast<const int4*>(sGc + jb);
            const int cjs[4] = {cj.x, cj.y, cj.z, cj.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int j = jb + e;
              float sv = s[kb][rg * 4 + e];
              if (a.rel_mode) {
                float bias;
                if (tile_grid) bias = q_grid ? sTbl[ciR + cjs[e]] : relx1;
                else bias = q_grid ? relx0 : ((j < a.S && qvalid) ? rel1d[ti - (j - a.P)] : 0.f);
                sv += bias;
              }
              if (a.dense && j < a.S) sv += a.dense[((long long)h * a.T + qrow) * a.dense_ld + j];
              bool masked = j >= a.S;
              if (a.causal) {
                if (tile_grid) masked |= (qi >= a.P) || (j > qi);
                else masked |= (qi >= a.P) && (j > qi);
              }
              sv = masked ? NEG_INF : sv;
              s[kb][rg * 4 + e] = sv;
              mx = fmaxf(mx, sv);
            }
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * LOG2E;     // m_run / l_run live in the exp2 domain
      if (__builtin_amdgcn_ballot_w64(mx > m_run + LAZY_MAX_SLACK)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == NEG_INF) ? 0.f : m_new));   // m_run = -inf -> 0
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
      }
      const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
      float ps4[4] = {0.f, 0.f, 0.f, 0.f};
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          U128 u;
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(s[kb][s2 * 8 + e], LOG2E, -m_use));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(s[kb][s2 * 8 + e + 1], LOG2E, -m_use));
            ps4[(e >> 1) & 3] += p0 + p1;
            u.w[e >> 1] = pack2bf(p0, p1);
          }
          pf[kb][s2] = u.b;
        }
      }
      l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
      // ---- O^T += V^T P^T ; slot (kh, e) <-> key kb*32 + 16*s2 + 4*kh + (e&3) + 8*(e>>2)
      const int i16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int colb = (db * 32 + g16 * 16 + (i16 & 3) * 4) * 2;
            const int r0 = kb * 32 + 16 * s2 + 4 * half + (i16 >> 2);
            U64 x, y;
            x.s = lds_read_tr(sVb(cur) + vx_off(r0, colb));
            y.s = lds_read_tr(sVb(cur) + vx_off(r0 + 8, colb));
            U128 vf; vf.w[0] = x.w[0]; vf.w[1] = x.w[1]; vf.w[2] = y.w[0]; vf.w[3] = y.w[1];
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.b, pf[kb][s2], oacc[db], 0, 0, 0);
          }
        }
      }
    }
    if (it + 1 < nsched) store_kv(cur ^ 1);
    __syncthreads();
  }

  // ---- finalize: lane (q, half) holds O[q][d = db*32 + (r&3) + 8*(r>>2) + 4*half]
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * (a.gain ? a.gain[h] : 1.f);
  {
    bf16_t* op = a.o + (long long)b * a.o_bs + (long long)qrow * a.ldo + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) store_tile_bf16(op + db * 32, oacc[db], inv, half, qvalid);
  }
  if (qvalid) {
    // log2-domain log-sum-exp (natural lse x log2 e): what the backward's exp2 consumes directly
    if (half == 0) a.lse[((long long)b * a.H + h) * a.T + qi] = m_run + __log2f(l_tot);
  }
}



// Rotation of a register inside each 32-lane half by a compile-time amount (ds_swizzle_b32, rotate mode: offset
// 0xC000 | dir << 10 | n << 5): lane x receives lane (x + N) & 31 of its half.  No address VGPR, no index arithmetic.
#ifndef DKV_ROT_DIR
#define DKV_ROT_DIR 0
#endif
template <int N>
__device__ __forceinline__ float rot32(float v) {
  if constexpr (N == 0) return v;
  else return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0xC000 | (DKV_ROT_DIR << 10) | (N << 5)));
}


__device__ __forceinline__ uint4 scale_bf16x8(uint4 v, float f) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    w[i] = pack2bf(__uint_as_float(w[i] << 16) * f, __uint_as_float(w[i] & 0xffff0000u) * f);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// (dK/dV kernel, grids of any width that is a multiple of 8 and >= 32)  One rotated dS register h -- rotation amount C = c_r,
// so lane x holds the term of key lane (x + C) & 31 -- added to the four class sums of its query set:
//   T every lane;  W lanes whose term wrapped around the 32-lane half (x + C >= 32: a compile-time lane interval);
//   K lanes whose term belongs to a key in the wave's SECOND grid row (bit (x + C) & 31 of `mk`, i.e. mk rotated right by C);
//   WK both.  The masks are applied as EXEC masks, the same for both halves of the wave.
template <int C>
__device__ __forceinline__ void seg_add(float h, unsigned mk, float& T, float& W, float& K, float& WK) {
  unsigned long long sv;
  if constexpr (C == 0) {
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_add_f32 %[T], %[T], %[h]\n\t"
        "s_mov_b32 exec_lo, %[mk]\n\ts_mov_b32 exec_hi, %[mk]\n\tv_add_f32 %[K], %[K], %[h]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [T] "+v"(T), [K] "+v"(K), [sv] "=&s"(sv)
        : [h] "v"(h), [mk] "s"(mk));
  } else {
    constexpr unsigned WM = 0xFFFFFFFFu << (32 - C);
    unsigned t0, t1;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_add_f32 %[T], %[T], %[h]\n\t"
        "s_mov_b32 exec_lo, %[wm]\n\ts_mov_b32 exec_hi, %[wm]\n\tv_add_f32 %[W], %[W], %[h]\n\t"
        "s_lshr_b32 %[t0], %[mk], %[c]\n\ts_lshl_b32 %[t1], %[mk], %[ci]\n\ts_or_b32 %[t0], %[t0], %[t1]\n\t"
        "s_mov_b32 exec_lo, %[t0]\n\ts_mov_b32 exec_hi, %[t0]\n\tv_add_f32 %[K], %[K], %[h]\n\t"
        "s_and_b32 %[t0], %[t0], %[wm]\n\t"
        "s_mov_b32 exec_lo, %[t0]\n\ts_mov_b32 exec_hi, %[t0]\n\tv_add_f32 %[WK], %[WK], %[h]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [T] "+v"(T), [W] "+v"(W), [K] "+v"(K), [WK] "+v"(WK), [sv] "=&s"(sv), [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [h] "v"(h), [mk] "s"(mk), [wm] "n"((int)WM), [c] "n"(C), [ci] "n"(32 - C)
        : "scc");
  }
}
// class sums of the rotated registers R .. R1-1 (rotation amount of register r: (r & 3) + 8 (r >> 2))
template <int R, int R1>
__device__ __forceinline__ void seg_sums_rec(const float (&h)[16], unsigned mk, float& T, float& W, float& K, float& WK) {
  if constexpr (R < R1) {
    seg_add<(R & 3) + 8 * (R >> 2)>(h[R], mk, T, W, K, WK);
    seg_sums_rec<R + 1, R1>(h, mk, T, W, K, WK);
  }
}
template <int R0, int R1>
__device__ __forceinline__ void seg_sums(const float (&h)[16], unsigned mk, float (&o)[4]) {
  float T = 0.f, W = 0.f, K = 0.f, WK = 0.f;
  seg_sums_rec<R0, R1>(h, mk, T, W, K, WK);
  o[0] = T; o[1] = W; o[2] = K; o[3] = WK;
}

// ---------------------------------------------------------------- backward
// Shared element logic: bias value for (query i, key j) is needed only to
// recompute P; the gradient of the bias goes to LDS histograms.
//
// (1) dK/dV kernel: one workgroup owns 128 keys (wave = 32 keys, lane = key) and
//     streams 32-query blocks of Q_ext / dO / lse / delta through two LDS stages by LDS-DMA.
//       S[q,key]  = Q_ext K_ext^T          (A = Q tile b128 reads, B = K regs; on a 32-wide grid the accumulator
//                                           is seeded with the rel-pos bias)
//       dP[q,key] = dO V^T                 (A = dO tile,           B = V rows of the workgroup, LDS)
//       dS = P (gain*dP - delta)
//       dV^T[d,key]   += dO^T P            (A = dO tile tr-reads,  B = P regs)
//       dK^T[c,key]   += Q_ext^T dS        (A = Q tile tr-reads,   B = dS regs)
//     so the key stays in the lane for S, P, dS and both accumulators.
// NW = waves per workgroup (4 or 8): 32 NW keys per workgroup.  8 waves when the per-head tables make one 4-wave workgroup
// exceed half of a CU's LDS (40 x 40 grids: 112 KiB) -- the 8 waves then SHARE one copy of the tables, two waves per SIMD
// instead of one (the staging of the query blocks is done by the first four waves).
template <bool HAS_POS, int NW>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, const int blk) {
  constexpr int NT = NW * 64;                     // threads
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // two staging buffers of DKV_STAGE_BYTES (one 32-query block each) open the LDS image
  const int n2dp = (a.n2d + 3) & ~3, n1d = a.rel_mode ? 2 * a.Lt - 1 : 0, n1dp = (n1d + 3) & ~3;
  unsigned char* sVk = smem + KT_BYTES + VT_BYTES + 512;                       // V rows of this WG's 32 NW keys
  float* sTbl = reinterpret_cast<float*>(sVk + (NW / 2) * VT_BYTES);           // rel2d[h]
  float* sHist = sTbl + n2dp;                                                  // d rel2d[h]
  // token-offset histogram and relx0 / relx1 accumulators: ONE COPY PER WAVE, summed in wave order at the end -- waves
  // adding to shared bins with LDS float atomics do so in an order that depends on their relative timing, and float
  // addition is not associative: the gradients then differ in the last bit from run to run
  float* sHist1 = sHist + n2dp;       // [NW][n1dp]
  float* sX = sHist1 + NW * n1dp;     // [NW][2]
  int* sGc = reinterpret_cast<int*>(sX + 2 * NW);
  // general grid width: per-block exchange area of the rel-pos gradient terms, [2 buffers][4 waves][header 4 | 3 x 64]
  constexpr int XW = 4 + 3 * 64;
  float* sXc = reinterpret_cast<float*>(sGc + ((a.P + 3) & ~3));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int nkt = (a.S + 32 * NW - 1) / (32 * NW);
  int bid = xcd_remap(blk, nkt * a.H * a.B);
  if (a.causal) {     // the tail tile (keys every query sees) and the early key tiles are the long workgroups: first
    int rank, bh;
    causal_order(blk, a.H * a.B, &rank, &bh);
    bid = bh * nkt + (rank + nkt - 1) % nkt;
  }
  const int kt = bid % nkt, h = (bid / nkt) % a.H, b = bid / (nkt * a.H);
  const int k0 = kt * 32 * NW;
  const int kw = __builtin_amdgcn_readfirstlane(k0 + wave * 32);
  const int kj = kw + (lane & 31);
  const bool kvalid = kj < a.S;
  const int krow = kvalid ? kj : a.S - 1;
  constexpr int NKS = HAS_POS ? 8 : 4;
  const float gain = a.gain ? a.gain[h] : 1.f;

  bf16x8 kf[NKS];
  {
    const bf16_t* kp = a.k + (long long)b * a.k_bs + (long long)krow * a.ldk + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(kp + ks * 16); kf[ks] = u.b; }
    // V rows of the workgroup's keys live in LDS (frees 16 VGPRs/lane so two workgroups fit a CU)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 3) + (NT / 8) * i, c = tid & 7, j = k0 + r;
      uint4 v4 = make_uint4(0, 0, 0, 0);
      if (j < a.S) v4 = *reinterpret_cast<const uint4*>(a.v + (long long)b * a.v_bs + (long long)j * a.ldv + h * 64 + c * 8);
      // stored as -gain * V: with the dP accumulator seeded with delta the MFMAs leave delta - gain * dP = -(dS / P)
      *reinterpret_cast<uint4*>(sVk + vx_off(r, c * 16)) = scale_bf16x8(v4, -gain);
    }
    if constexpr (HAS_POS) {
      const bf16_t* pp = a.pk + (long long)krow * a.ldpk + h * 64 + half * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(pp + ks * 16); kf[4 + ks] = u.b; }
    }
  }
  if (a.rel_mode) {
    stage_table<float, false, NT>(sTbl, a.rel2d + (long long)h * a.n2d, a.n2d, tid);
    stage_table<int, false, NT>(sGc, a.gcode, a.P, tid);
    for (int i = tid; i < a.n2d; i += NT) sHist[i] = 0.f;
    for (int i = tid; i < NW * n1dp + 2 * NW; i += NT) sHist1[i] = 0.f;
  }
  const bool k_grid = kj < a.P;
  const bool wave_kgrid = kw + 31 < a.P;
  const int cj = (a.rel_mode && k_grid) ? a.gcode[kj] - a.code_bias : 0;   // table index = ci - cj
  const int tj = kj - a.P;
  const float* rel1d = a.rel_mode ? a.rel1d + (long long)h * n1d + (a.Lt - 1) : nullptr;
  const float relx0 = a.rel_mode ? a.relx[h * 2 + 0] : 0.f, relx1 = a.rel_mode ? a.relx[h * 2 + 1] : 0.f;
  float gx0 = 0.f, gx1 = 0.f;
  const bool row32 = a.rel_mode && a.grid_w == 32;
  const int cj0 = __builtin_amdgcn_readfirstlane(cj);     // code of the wave's first key (x = 0 when row32)
  const int xl = (lane & 31) + 4 * half;
  // Grids of any other width that is a multiple of 8 (>= 32; BASELINE configs[3]: 40): the wave's 32 keys span at most two
  // grid rows, a block's 32 queries likewise, and every aligned group of 8 lies in one row.  `mk`: key lanes in the wave's
  // second row (a code jumps by w instead of 1 at a row end).  Seeds: one table address per group of 8 queries.  Gradient:
  // the same in-register rotation as on a 32-wide grid, the rotated terms summed per (query row, key row, wrap) class
  // (seg_sums) -- three bins per lane and block, handed through `sXc` to ONE wave per block that adds them to the table in
  // a fixed order: bit-reproducible, no LDS float atomics (the 4 waves share table rows here).
  const bool rowseg = a.rel_mode && !row32 && a.grid_w >= 32 && (a.grid_w & 7) == 0;
  const unsigned mk = rowseg ? (unsigned)__builtin_amdgcn_ballot_w64(k_grid && (cj - cj0 != (lane & 31))) : 0u;
  if (rowseg) {       // (the exchange area exists only on these grids: the host sizes the LDS accordingly)
    for (int i = tid; i < 2 * NW * XW; i += NT) sXc[i] = 0.f;
  }
  auto seg_duty = [&](int buf) {
#pragma unroll 1
    for (int w = 0; w < NW; ++w) {
      const float* xs = sXc + (buf * NW + w) * XW;
      if (__builtin_amdgcn_readfirstlane(__float_as_int(xs[0])) == 0) continue;
      const int bq = __builtin_amdgcn_readfirstlane(__float_as_int(xs[1])) + (lane < 32 ? -lane : 64 - lane);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int bin = bq + (d - 1) * (a.grid_w - 1);
        const float v = xs[4 + d * 64 + lane];
        // the lanes of one instruction address distinct bins; steps follow each other in program order on ONE wave
        if (lane != 32 && bin >= 0 && bin < a.n2d) sHist[bin] += v;
      }
    }
  };

  // ---- q-tile schedule
  const int nqt = (a.T + 63) >> 6;
  int qs = 0, qe = nqt;
  // causal: a pure-grid key tile is visible only to grid queries at or after it; a tile holding tail
  // keys (visible to every grid query) keeps the full range and relies on the per-wave skip below
  if (a.causal && k0 + 32 * NW - 1 < a.P) { qs = k0 >> 6; qe = a.P >> 6; }
  const int nsched = qe - qs;

  const bf16_t* qb_ = a.q + (long long)b * a.q_bs + h * 64;
  const bf16_t* pqb_ = HAS_POS ? a.pq + h * 64 : nullptr;
  const bf16_t* dob_ = a.dO + (long long)b * a.do_bs + h * 64;
  const float* lseb = a.lse + ((long long)b * a.H + h) * a.T;
  const float* delb = a.delta + ((long long)b * a.H + h) * a.T;
  // ---- staging by LDS-DMA, two stages of one 32-query block each:
  //   [Q_ext 32 x 256 B | dO 32 x 128 B | lse 32 f32 | delta 32 f32]
  // wave w moves Q rows 8w..8w+7 (two 1-KiB pieces), dO rows 8w..8w+7 (one piece); wave 0 also the
  // statistics.  Rows past T are clamped to T-1 (their P is masked to zero below).
  constexpr int STG = DKV_STAGE_BYTES;
  const unsigned lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int ib, int st) {
    const unsigned base = lds0 + st * STG;
    if (NW > 4 && wv >= 4) return;          // (8-wave workgroups: the first four waves stage the query block)
    // lane constants derived from a re-materialised lane id on every call (six VGPRs less across the loop)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    int q_row[2], q_c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      q_row[i] = (wv * 2 + i) * 4 + (ln >> 4);
      q_c[i] = (ln & 15) ^ (((q_row[i] & 3) << 2) | ((q_row[i] >> 2) & 3));
    }
    const int o_row = wv * 8 + (ln >> 3);
    const int o_c = ((ln & 7) ^ ((((o_row >> 1) & 1) << 2) | ((o_row >> 2) & 3))) & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // wave-uniform bases + 32-bit lane offsets (a row of one batch element is < 2^31 bytes away)
      const int qr = min(ib + q_row[i], a.T - 1);
      if (q_c[i] < 8) lds_dma16_gs(qb_, (qr * a.ldq + q_c[i] * 8) * 2, base + (wv * 2 + i) * 1024);
      else if (HAS_POS) lds_dma16_gs(pqb_, (qr * a.ldpq + (q_c[i] - 8) * 8) * 2, base + (wv * 2 + i) * 1024);
    }
    {
      const int qr = min(ib + o_row, a.T - 1);
      lds_dma16_gs(dob_, (qr * a.lddo + o_c * 8) * 2, base + 8192 + wv * 1024);
    }
    if (wv == 0) {
      // lanes 0..31: lse, lanes 32..63: delta (LDS-DMA places lane i at base + 4*i whichever lanes are active)
      const int qr = min(ib + (ln & 31), a.T - 1);
      if (ln < 32) lds_dma4_gs(lseb, qr * 4, base + 8192 + 4096);
      else lds_dma4_gs(delb, qr * 4, base + 8192 + 4096);
    }
  };

  f32x16 dv[2], dk[NKS / 2];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    dv[0][e] = 0.f; dv[1][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NKS / 2; ++c) dk[c][e] = 0.f;
  }
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  // Per-lane LDS offsets of the block's ~40 operand reads.  The swizzles are XORs on the chunk bits, so every read is
  // one of five bases XOR / plus a literal:
  //   Q rows   (S):   aQ ^ (ks << 5)                 dO rows (dP): aO ^ (ks << 5)       V rows: aV ^ (ks << 5)
  //   dO^T (dV): ((aOt ^ (db << 6)) + s2 * 2048), second row group (+ 1024) ^ 32
  //   Q^T  (dK): ((aQt ^ (cb << 6)) + s2 * 4096), second row group (+ 2048) ^ 32
  const int nblk = min(qe * 2, (a.T + 31) >> 5) - qs * 2;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) consume_frag(kf[ks]);
  if (nblk > 0) issue(qs * 64, 0);
  for (int n = 0; n < nblk; ++n) {
    const int ib = (qs * 2 + n) * 32;
    const unsigned char* sQ = smem + (n & 1) * STG;
    const unsigned char* sO = sQ + 8192;
    const float* sL = reinterpret_cast<const float*>(sQ + 8192 + 4096);
    lds_dma_wait();
    __syncthreads();              // block n has landed; everyone is done with block n-1 (and the table init)
    if (n + 1 < nblk) issue(ib + 32, (n + 1) & 1);
    if (rowseg) {
      if (n > 0 && wv == ((n - 1) & (NW - 1))) seg_duty((n - 1) & 1);       // the terms of block n-1, by one wave
      if (lane == 0) sXc[((n & 1) * NW + wv) * XW] = __int_as_float(0);   // this wave's slot of block n: empty so far
    }
    {
      const bool skip = (kw >= a.S) || (a.causal && wave_kgrid && ((ib + 31 < kw) || (ib >= a.P)));
      if (skip) continue;
      const bool qb_grid = ib + 31 < a.P;
      const int fast = (a.rel_mode && qb_grid && wave_kgrid) ? 1 : 0;
      // block classes with a straight-line element body (everything else -- text x text, grids that are not 32 wide,
      // causal layers without a bias -- takes the general per-element path below):
      //   1 grid x grid on a 32-wide grid: bias from the delta table seeds S, gradient by diagonal sums
      //   2 grid queries x text keys / text queries x grid keys: the bias is ONE scalar per head (relx[0] / relx[1]),
      //     its gradient the sum of dS.  P % 32 == 0 and 32-key waves: a block lies entirely on one side.
      //   3 no relative bias at all (cross attention), not causal
      const bool qb_text = ib >= a.P;
      //   4 grid x grid on a grid of another width that is a multiple of 8 (see `rowseg` above)
      const int cls = (row32 && fast == 1) ? 1
                    : (rowseg && fast == 1) ? 4
                    : (a.rel_mode && ((qb_grid && kw >= a.P) || (qb_text && wave_kgrid))) ? 2
                    : (!a.rel_mode && !a.causal) ? 3 : 0;
      const bool edge = (ib + 32 > a.T) || (kw + 32 > a.S);      // rows / keys past the end: masked element-wise
      // The five bases (and the lane's x coordinate for the histogram) are re-materialised per block: as plain loop
      // invariants the compiler hoists all ~40 derived addresses and 16 shuffle indices into VGPRs, spills part of
      // them, and every scratch reload waits with vmcnt(0) -- i.e. for the NEXT block's LDS-DMA -- in the loop.
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int half_i = ln >> 5, colT = ((ln >> 4) & 1) * 16 + (ln & 3) * 4, rT = 4 * half_i + ((ln & 15) >> 2);
      const int bQ = kx_off(ln & 31, half_i), bO = vx_off(ln & 31, half_i * 16), bV = vx_off(wv * 32 + (ln & 31), half_i * 16);
      const int bOt = vx_off(rT, colT * 2), bQt = kx_off(rT, colT >> 3) + (colT & 7) * 2;
      f32x16 s, dp;
      // dP accumulator seeded with delta (V rows are stored as -gain * V): dS = -P * dp
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 d4 = *reinterpret_cast<const float4*>(sL + 32 + 8 * rg + 4 * half_i);
        dp[rg * 4] = d4.x; dp[rg * 4 + 1] = d4.y; dp[rg * 4 + 2] = d4.z; dp[rg * 4 + 3] = d4.w;
      }
      const bool seeded = cls == 1;
      int hidx_rw = 0, seg_c0 = 0, seg_nq0 = 4;
      float hold = 0.f;
      if (seeded) {
        // the block's queries are one grid row: the bias of element r sits at a constant offset from one address
        // and seeds the accumulator (natural units)
        const int code_i = sGc[ib];
        const float* tp = sTbl + (code_i - cj + 4 * half);
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = tp[(e & 3) + 8 * (e >> 2)];
        // histogram bin this lane updates at the end of the block: (code_i - cj0) is dx = 0 of this (query row, key row)
        hidx_rw = code_i - cj0 + (lane < 32 ? -lane : 64 - lane);
        hold = sHist[lane != 32 ? hidx_rw : 0];
      } else if (cls == 4) {
        // one table address per group of 8 queries; nq0 = groups in the block's first grid row
        int cg[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) cg[rg] = sGc[ib + 8 * rg];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float* tp = sTbl + (cg[rg] - cj + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[rg * 4 + e] = tp[e];
        }
        seg_c0 = __builtin_amdgcn_readfirstlane(cg[0]);
        seg_nq0 = 1 + (cg[1] - cg[0] == 8) + (cg[2] - cg[0] == 16) + (cg[3] - cg[0] == 24);
        seg_nq0 = __builtin_amdgcn_readfirstlane(seg_nq0);
      } else {
        const float c0 = cls == 2 ? (qb_grid ? relx0 : relx1) : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = c0;
      }
      {
        // all Q fragments are requested before the first MFMA, the dO / V fragments travel while the S MFMAs run:
        // two exposed LDS latencies per block instead of one per MFMA
        bf16x8 qf[NKS], of[4], vfr[4];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = lds_read_b128(sQ + (bQ ^ (ks << 5)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], s, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          of[ks] = lds_read_b128(sO + (bO ^ (ks << 5)));
          vfr[ks] = lds_read_b128(sVk + (bV ^ (ks << 5)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[ks], vfr[ks], dp, 0, 0, 0);
      }
      // element r <-> query ib + (r&3) + 8*(r>>2) + 4*half ; key = kj (lane)
      bf16x8 pfr[2], dsf[2];
      float hval[16];      // seeded blocks: dS rotated onto its histogram lane (summed after the dV / dK MFMAs)
      if (cls) {
        // FLAGS: 1 diagonal-sum histogram (class 1), 2 causal mask, 4 row / key validity masks, 8 sum of dS (class 2)
        auto body = [&](auto flags_tag) {
          constexpr int FLAGS = decltype(flags_tag)::value;
          float dsr[16];
          U128 up[2], ud[2];
          const int rowlim = a.T - ib;             // FLAGS & 4: query row il + e is real iff il + e < rowlim
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int il = 8 * rg + 4 * half;
            const float4 l4 = *reinterpret_cast<const float4*>(sL + il);
            const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
            const int di = kj - (ib + il);         // masked (causal) iff kj > i  <=>  di > e
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float p = __builtin_amdgcn_exp2f(fmaf(s[rg * 4 + e], LOG2E, -ls[e]));
              if (FLAGS & 2) p = (di > e) ? 0.f : p;
              if (FLAGS & 4) p = (kvalid && il + e < rowlim) ? p : 0.f;
              pv[e] = p;
              dsr[rg * 4 + e] = -p * dp[rg * 4 + e];
            }
            up[rg >> 1].w[(rg & 1) * 2] = pack2bf(pv[0], pv[1]); up[rg >> 1].w[(rg & 1) * 2 + 1] = pack2bf(pv[2], pv[3]);
            ud[rg >> 1].w[(rg & 1) * 2] = pack2bf(dsr[rg * 4], dsr[rg * 4 + 1]);
            ud[rg >> 1].w[(rg & 1) * 2 + 1] = pack2bf(dsr[rg * 4 + 2], dsr[rg * 4 + 3]);
          }
          pfr[0] = up[0].b; pfr[1] = up[1].b; dsf[0] = ud[0].b; dsf[1] = ud[1].b;
          if constexpr (FLAGS & 1) {
            // d rel2d: this wave's keys are one grid row (x_j = lane & 31) and the block's queries another
            // (x_i = c_r + 4 * half, c_r = (r&3) + 8*(r>>2)).  Register r is rotated inside each half by c_r (an
            // immediate of ds_swizzle): lane (half, x) then holds the term of bin dx = 4 * half - x (no wrap,
            // x + c_r <= 31) or dx = 4 * half - x + 32 (wrap).  The 16 rotated registers are summed after the dV / dK
            // MFMAs (all 16 swizzles in flight at once), then 2 half-wave LDS adds per block (one LDS float atomic
            // per element instead more than doubles the kernel time).
            hval[0] = rot32<0>(dsr[0]);   hval[1] = rot32<1>(dsr[1]);   hval[2] = rot32<2>(dsr[2]);   hval[3] = rot32<3>(dsr[3]);
            hval[4] = rot32<8>(dsr[4]);   hval[5] = rot32<9>(dsr[5]);   hval[6] = rot32<10>(dsr[6]);  hval[7] = rot32<11>(dsr[7]);
            hval[8] = rot32<16>(dsr[8]);  hval[9] = rot32<17>(dsr[9]);  hval[10] = rot32<18>(dsr[10]); hval[11] = rot32<19>(dsr[11]);
            hval[12] = rot32<24>(dsr[12]); hval[13] = rot32<25>(dsr[13]); hval[14] = rot32<26>(dsr[14]); hval[15] = rot32<27>(dsr[15]);
          }
          if constexpr (FLAGS & 8) {
            float g = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) g += dsr[e];
            if (qb_grid) gx0 += g; else gx1 += g;
          }
        };
        if (cls == 1 || cls == 4) {
          // only blocks crossing the diagonal hold masked elements (blocks entirely above it were skipped)
          if (a.causal && kw + 31 > ib) body(std::integral_constant<int, 3>{}); else body(std::integral_constant<int, 1>{});
        } else if (cls == 2) {
          body(std::integral_constant<int, 12>{});
        } else {
          if (edge) body(std::integral_constant<int, 4>{}); else body(std::integral_constant<int, 0>{});
        }
      } else
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        U128 up, ud;
#pragma unroll
        for (int rg2 = 0; rg2 < 2; ++rg2) {
          const int rg = s2 * 2 + rg2;
          const int iq = ib + 8 * rg + 4 * half, il = 8 * rg + 4 * half;
          const float4 l4 = *reinterpret_cast<const float4*>(sL + il);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
          float pv[4], dsv[4];
          if (fast == 1) {
            const int4 c4 = *reinterpret_cast<const int4*>(sGc + iq);
            const int cis[4] = {c4.x, c4.y, c4.z, c4.w};
            const int di = kj - iq;                // masked (causal) iff kj > i  <=>  di > e
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int hidx = cis[e] - cj;
              float p = __builtin_amdgcn_exp2f(fmaf(s[rg * 4 + e] + sTbl[hidx], LOG2E, -ls[e]));
              if (a.causal) p = (di > e) ? 0.f : p;
              const float ds = -p * dp[rg * 4 + e];
              pv[e] = p; dsv[e] = ds;
              atomicAdd(&sHist[hidx], ds);
            }
          } else {
            int4 c4 = make_int4(0, 0, 0, 0);
            const bool qg = ib < a.P;            // P % 32 == 0: whole block on one side
            if (a.rel_mode && qg) c4 = *reinterpret_cast<const int4*>(sGc + iq);
            const int cis[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = iq + e;
              float sv = s[rg * 4 + e];
              int hidx = 0;
              if (a.rel_mode) {
                float bias;
                if (qg) {
                  if (k_grid) { hidx = cis[e] - cj; bias = sTbl[hidx]; }
                  else bias = relx0;
                } else {
                  if (k_grid) bias = relx1;
                  else {
                    const bool ok = kvalid && i < a.T;
                    hidx = ok ? (i - a.P) - tj + a.Lt - 1 : 0;
                    bias = ok ? rel1d[(i - a.P) - tj] : 0.f;
                  }
                }
                sv += bias;
              }
              bool masked = !kvalid || i >= a.T;
              if (a.causal) {
                if (k_grid) masked |= (i >= a.P) || (kj > i);
                else masked |= (i >= a.P) && (kj > i);
              }
              const float p = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(sv, LOG2E, -ls[e]));
              const float ds = -p * dp[rg * 4 + e];
              pv[e] = p; dsv[e] = ds;
              if (a.rel_mode) {
                if (qg) { if (k_grid) atomicAdd(&sHist[hidx], ds); else gx0 += ds; }
                else { if (k_grid) gx1 += ds; else if (kvalid && i < a.T) atomicAdd(&sHist1[wv * n1dp + hidx], ds); }
              }
            }
          }
          up.w[rg2 * 2] = pack2bf(pv[0], pv[1]); up.w[rg2 * 2 + 1] = pack2bf(pv[2], pv[3]);
          ud.w[rg2 * 2] = pack2bf(dsv[0], dsv[1]); ud.w[rg2 * 2 + 1] = pack2bf(dsv[2], dsv[3]);
        }
        pfr[s2] = up.b; dsf[s2] = ud.b;
      }
      // dV^T += dO^T P ; dK^T += Q^T dS ; slot (kh,e) <-> query ib + 16*s2 + 4*kh + (e&3) + 8*(e>>2)
      // all transposed operand reads of one s2 half are issued before its MFMAs (the scheduling barrier keeps
      // the compiler from sinking each read next to its MFMA, which serialises LDS latency 6 times per half;
      // batching both halves at once spills)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // rows 16*s2 + 4*half + (i16>>2) (+ 8), columns db*32 / cb*32 + g16*16 + (i16&3)*4 of the stage
        U128 fo[2], fq[NKS / 2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int o0 = (bOt ^ (db << 6)) + s2 * 2048;
          U64 x, y;
          x.s = lds_read_tr(sO + o0);
          y.s = lds_read_tr(sO + ((o0 + 1024) ^ 32));
          fo[db].w[0] = x.w[0]; fo[db].w[1] = x.w[1]; fo[db].w[2] = y.w[0]; fo[db].w[3] = y.w[1];
        }
#pragma unroll
        for (int cb = 0; cb < NKS / 2; ++cb) {
          const int o0 = (bQt ^ (cb << 6)) + s2 * 4096;
          U64 x, y;
          x.s = lds_read_tr(sQ + o0);
          y.s = lds_read_tr(sQ + ((o0 + 2048) ^ 32));
          fq[cb].w[0] = x.w[0]; fq[cb].w[1] = x.w[1]; fq[cb].w[2] = y.w[0]; fq[cb].w[3] = y.w[1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 2; ++db) dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[db].b, pfr[s2], dv[db], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NKS / 2; ++cb) dk[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[cb].b, dsf[s2], dk[cb], 0, 0, 0);
      }
      if (seeded) {
        // accT: every term; accA: the terms without wrap, x + c_r <= 31 -- a compile-time lane interval per
        // register, applied as an EXEC mask (one masked add instead of compare + select + add)
        float accT, accA;
        unsigned long long sv_exec;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "v_add_f32 %[t], %[h0], %[h1]\n\t"
            "v_add_f32 %[a], %[h2], %[h3]\n\t"
            "v_add_f32 %[t], %[t], %[h4]\n\t"
            "v_add_f32 %[a], %[a], %[h5]\n\t"
            "v_add_f32 %[t], %[t], %[h6]\n\t"
            "v_add_f32 %[a], %[a], %[h7]\n\t"
            "v_add_f32 %[t], %[t], %[h8]\n\t"
            "v_add_f32 %[a], %[a], %[h9]\n\t"
            "v_add_f32 %[t], %[t], %[h10]\n\t"
            "v_add_f32 %[a], %[a], %[h11]\n\t"
            "v_add_f32 %[t], %[t], %[h12]\n\t"
            "v_add_f32 %[a], %[a], %[h13]\n\t"
            "v_add_f32 %[t], %[t], %[h14]\n\t"
            "v_add_f32 %[a], %[a], %[h15]\n\t"
            "v_add_f32 %[t], %[t], %[a]\n\t"
            "v_mov_b32 %[a], %[h0]\n\t"
            "s_mov_b32 exec_lo, 0x7fffffff\n\ts_mov_b32 exec_hi, 0x7fffffff\n\tv_add_f32 %[a], %[a], %[h1]\n\t"
            "s_mov_b32 exec_lo, 0x3fffffff\n\ts_mov_b32 exec_hi, 0x3fffffff\n\tv_add_f32 %[a], %[a], %[h2]\n\t"
            "s_mov_b32 exec_lo, 0x1fffffff\n\ts_mov_b32 exec_hi, 0x1fffffff\n\tv_add_f32 %[a], %[a], %[h3]\n\t"
            "s_mov_b32 exec_lo, 0x00ffffff\n\ts_mov_b32 exec_hi, 0x00ffffff\n\tv_add_f32 %[a], %[a], %[h4]\n\t"
            "s_mov_b32 exec_lo, 0x007fffff\n\ts_mov_b32 exec_hi, 0x007fffff\n\tv_add_f32 %[a], %[a], %[h5]\n\t"
            "s_mov_b32 exec_lo, 0x003fffff\n\ts_mov_b32 exec_hi, 0x003fffff\n\tv_add_f32 %[a], %[a], %[h6]\n\t"
            "s_mov_b32 exec_lo, 0x001fffff\n\ts_mov_b32 exec_hi, 0x001fffff\n\tv_add_f32 %[a], %[a], %[h7]\n\t"
            "s_mov_b32 exec_lo, 0x0000ffff\n\ts_mov_b32 exec_hi, 0x0000ffff\n\tv_add_f32 %[a], %[a], %[h8]\n\t"
            "s_mov_b32 exec_lo, 0x00007fff\n\ts_mov_b32 exec_hi, 0x00007fff\n\tv_add_f32 %[a], %[a], %[h9]\n\t"
            "s_mov_b32 exec_lo, 0x00003fff\n\ts_mov_b32 exec_hi, 0x00003fff\n\tv_add_f32 %[a], %[a], %[h10]\n\t"
            "s_mov_b32 exec_lo, 0x00001fff\n\ts_mov_b32 exec_hi, 0x00001fff\n\tv_add_f32 %[a], %[a], %[h11]\n\t"
            "s_mov_b32 exec_lo, 0x000000ff\n\ts_mov_b32 exec_hi, 0x000000ff\n\tv_add_f32 %[a], %[a], %[h12]\n\t"
            "s_mov_b32 exec_lo, 0x0000007f\n\ts_mov_b32 exec_hi, 0x0000007f\n\tv_add_f32 %[a], %[a], %[h13]\n\t"
            "s_mov_b32 exec_lo, 0x0000003f\n\ts_mov_b32 exec_hi, 0x0000003f\n\tv_add_f32 %[a], %[a], %[h14]\n\t"
            "s_mov_b32 exec_lo, 0x0000001f\n\ts_mov_b32 exec_hi, 0x0000001f\n\tv_add_f32 %[a], %[a], %[h15]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [t] "=&v"(accT), [a] "=&v"(accA), [sv] "=&s"(sv_exec)
            : [h0] "v"(hval[0]), [h1] "v"(hval[1]), [h2] "v"(hval[2]), [h3] "v"(hval[3]), [h4] "v"(hval[4]), [h5] "v"(hval[5]),
              [h6] "v"(hval[6]), [h7] "v"(hval[7]), [h8] "v"(hval[8]), [h9] "v"(hval[9]), [h10] "v"(hval[10]),
              [h11] "v"(hval[11]), [h12] "v"(hval[12]), [h13] "v"(hval[13]), [h14] "v"(hval[14]), [h15] "v"(hval[15]));
        const float accB = accT - accA;
        // half 1 holds its bins 4 lanes further on (x_i = c_r + 4): bring bin -x / 32 - x to lane x of half 1 too
        const float ra = rot32<4>(accA), rb = rot32<4>(accB);
        const bool lo = lane < 32, in = (lane & 31) <= 27;
        const float a2 = lo ? accA : (in ? ra : 0.f), b2 = lo ? accB : (in ? rb : ra);
        // halves combined in one VALU op: lanes 0..31 end up with the dx = -u total, lanes 32..63 with dx = 32 - u
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(b2), false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        // plain read-modify-write: between two barriers the four waves work on the same query row and four different
        // key rows, i.e. on four different rows of the table (an LDS float atomic costs ~30 us per layer here)
        if (lane != 32) sHist[hidx_rw] = hold + tot;
      } else if (cls == 4) {
        // class sums of the query groups in the block's first grid row (set 0) and in its second (set 1)
        float c0[4], c1[4];
        switch (seg_nq0) {
          case 1: seg_sums<0, 4>(hval, mk, c0); seg_sums<4, 16>(hval, mk, c1); break;
          case 2: seg_sums<0, 8>(hval, mk, c0); seg_sums<8, 16>(hval, mk, c1); break;
          case 3: seg_sums<0, 12>(hval, mk, c0); seg_sums<12, 16>(hval, mk, c1); break;
          default: seg_sums<0, 16>(hval, mk, c0); seg_sums<16, 16>(hval, mk, c1); break;
        }
        // {T, W, K, WK} -> (no wrap, wrap) sums per row offset d = [query in 2nd row] - [key in 2nd row]:
        //   set 0: first-row keys d = 0, second-row keys d = -1;  set 1: first-row keys d = +1, second-row keys d = 0
        const float nw[3] = {c0[2] - c0[3], (c0[0] - c0[1] - c0[2] + c0[3]) + (c1[2] - c1[3]), c1[0] - c1[1] - c1[2] + c1[3]};
        const float wr[3] = {c0[3], (c0[1] - c0[3]) + c1[3], c1[1] - c1[3]};
        float* xs = sXc + ((n & 1) * NW + wv) * XW;
        const bool lo = lane < 32, in = (lane & 31) <= 27;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          // (as on the 32-wide grid) half 1 holds its bins 4 lanes further on; lanes 0..31 end up with the dx = -u
          // totals, lanes 32..63 with the dx = 32 - u ones
          const float ra = rot32<4>(nw[d]), rb = rot32<4>(wr[d]);
          const float a2 = lo ? nw[d] : (in ? ra : 0.f), b2 = lo ? wr[d] : (in ? rb : ra);
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(b2), false, false);
          xs[4 + d * 64 + lane] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        if (lane == 0) { xs[1] = __int_as_float(seg_c0 - cj0); xs[0] = __int_as_float(1); }
      }
    }
  }
  if (rowseg && nblk > 0) {
    __syncthreads();
    if (wv == ((nblk - 1) & (NW - 1))) seg_duty((nblk - 1) & 1);
  }

  // ---- write dV, dK, dpos_k partial: lane = key, reg r <-> column (r&3) + 8*(r>>2) + 4*half
  if (kvalid) {
    bf16_t* dvp = a.dv + (long long)b * a.dv_bs + (long long)kj * a.lddv + h * 64;
    bf16_t* dkp = a.dk + (long long)b * a.dk_bs + (long long)kj * a.lddk + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      store_tile_bf16(dvp + db * 32, dv[db], gain, half, true);
      store_tile_bf16(dkp + db * 32, dk[db], 1.f, half, true);
    }
    if constexpr (HAS_POS) {
      bf16_t* pp = a.dpk + ((long long)b * a.S + kj) * (a.H * 64) + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db) store_tile_bf16(pp + db * 32, dk[2 + db], 1.f, half, true);
    }
  }
  if (a.rel_mode) {
    gx0 = warp_sum(gx0); gx1 = warp_sum(gx1);
    if (lane == 0) { sX[wv * 2] = gx0; sX[wv * 2 + 1] = gx1; }
    __syncthreads();
    // partial slots are per 128 keys (nparts = B * ceil(S / 128)): an 8-wave workgroup fills the first of its two slots
    // and clears the second
    const int np128 = (a.S + 127) >> 7;
    const int part = b * np128 + kt * (NW / 4);
    float* o2 = a.drel2d_part + ((long long)h * a.nparts + part) * a.n2d;
    for (int i = tid; i < a.n2d; i += NT) o2[i] = sHist[i];
    float* o1 = a.drel1d_part + ((long long)h * a.nparts + part) * n1d;
    for (int i = tid; i < n1d; i += NT) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += sHist1[w * n1dp + i];
      o1[i] = t;
    }
    if (tid < 2) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += sX[2 * w + tid];
      a.drelx_part[((long long)h * a.nparts + part) * 2 + tid] = t;
    }
    if (NW == 8 && kt * 2 + 1 < np128) {
      for (int i = tid; i < a.n2d; i += NT) o2[a.n2d + i] = 0.f;
      for (int i = tid; i < n1d; i += NT) o1[n1d + i] = 0.f;
      if (tid < 2) a.drelx_part[((long long)h * a.nparts + part + 1) * 2 + tid] = 0.f;
    }
  }
}

// (2) dQ kernel: same shape as the forward (lane = query):
//       S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (gain*dP^T - delta),
//       dQ_ext^T[c,q] += K_ext^T dS^T  (A = K tile tr-reads, B = dS regs)
template <bool HAS_POS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs a) { attn_bwd_dkv_body<HAS_POS, 4>(a, blockIdx.x); }
template <bool HAS_POS>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv8_kernel(AttnArgs a) { attn_bwd_dkv_body<HAS_POS, 8>(a, blockIdx.x); }

template <bool HAS_POS>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, const int blk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto sKb = [&](int buf) { return smem + buf * (KT_BYTES + VT_BYTES); };
  auto sVb = [&](int buf) { return smem + buf * (KT_BYTES + VT_BYTES) + KT_BYTES; };
  float* sTbl = reinterpret_cast<float*>(smem + 2 * (KT_BYTES + VT_BYTES));
  int* sGc = reinterpret_cast<int*>(sTbl + ((a.n2d + 3) & ~3));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int nq = (a.T + 127) >> 7;
  int qt, h, b;
  if (a.causal) {
    int rank, bh;
    causal_order(blk, a.H * a.B, &rank, &bh);
    qt = nq - 1 - rank; h = bh % a.H; b = bh / a.H;
  } else {
    const int bid = xcd_remap(blk, nq * a.H * a.B);
    qt = bid % nq; h = (bid / nq) % a.H; b = bid / (nq * a.H);
  }
  const int q0 = qt * 128, qw = q0 + wave * 32;
  const int qi = qw + (lane & 31);
  const bool qvalid = qi < a.T;
  const int qrow = qvalid ? qi : a.T - 1;
  constexpr int NKS = HAS_POS ? 8 : 4;
  const float gain = a.gain ? a.gain[h] : 1.f;

  bf16x8 qf[NKS], dof[4];
  {
    const bf16_t* qp = a.q + (long long)b * a.q_bs + (long long)qrow * a.ldq + h * 64 + half * 8;
    const bf16_t* op = a.dO + (long long)b * a.do_bs + (long long)qrow * a.lddo + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      U128 u; u.v = *reinterpret_cast<const uint4*>(qp + ks * 16); qf[ks] = u.b;
      U128 w; w.v = qvalid ? *reinterpret_cast<const uint4*>(op + ks * 16) : make_uint4(0, 0, 0, 0); dof[ks] = w.b;
    }
    if (HAS_POS) {
      const bf16_t* pp = a.pq + (long long)qrow * a.ldpq + h * 64 + half * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(pp + ks * 16); qf[4 + ks] = u.b; }
    }
  }
  const float nlse_q = qvalid ? -a.lse[((long long)b * a.H + h) * a.T + qi] : -INFINITY;   // log2 units
  const bool row32 = a.rel_mode && a.grid_w == 32;
  const bool rowseg = a.rel_mode && !row32 && a.grid_w >= 32 && (a.grid_w & 7) == 0;    // see the forward kernel
  const float del_q = qvalid ? a.delta[((long long)b * a.H + h) * a.T + qi] : 0.f;
  if (a.rel_mode) {
    // the table is kept REVERSED in LDS (sTbl[n2d - 1 - i] = rel2d[i]): a lane is a query and its registers are keys of
    // increasing x, i.e. of decreasing table index; reversed, the 16 seeds of a block are read in register order
    // (ds_read2_b32 pairs land in the accumulator tuple; in table order every seed needed a v_mov: ~60 per tile)
    stage_table<float, true>(sTbl, a.rel2d + (long long)h * a.n2d, a.n2d, tid);
    stage_table(sGc, a.gcode, a.P, tid);
  }
  const bool q_grid = a.rel_mode && qi < a.P;
  const int ciR = q_grid ? a.n2d - 1 - (a.gcode[qi] + a.code_bias) : 0;    // reversed-table index = ciR + cj
  const int ti = qi - a.P;
  const float relx0 = a.rel_mode ? a.relx[h * 2 + 0] : 0.f, relx1 = a.rel_mode ? a.relx[h * 2 + 1] : 0.f;
  const float* rel1d = a.rel_mode ? a.rel1d + (long long)h * (2 * a.Lt - 1) + (a.Lt - 1) : nullptr;

  const int ntile = (a.S + 63) >> 6;
  const int Pk = a.rel_mode || a.causal ? a.P : a.S;
  const int ngrid = Pk >> 6;
  int g_end = ngrid;
  if (a.causal) {
    int imax = (q0 < a.P) ? min(q0 + 127, a.P - 1) : -1;
    g_end = imax >= 0 ? min(ngrid, (imax >> 6) + 1) : 0;
  }
  const int nsched = g_end + (ntile - ngrid);
  auto tile_of = [&](int it) { return it < g_end ? it : ngrid + (it - g_end); };

  uint4 rk[4], rv[2];
  const bf16_t* kb_ = a.k + (long long)b * a.k_bs + h * 64;
  const bf16_t* vb_ = a.v + (long long)b * a.v_bs + h * 64;
  const bf16_t* pkb_ = HAS_POS ? a.pk + h * 64 : nullptr;
  auto load_kv = [&](int j0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7, j = j0 + r;
      rk[i] = make_uint4(0, 0, 0, 0); rk[2 + i] = make_uint4(0, 0, 0, 0); rv[i] = make_uint4(0, 0, 0, 0);
      if (j < a.S) {
        // wave-uniform base + 32-bit lane offset (saddr addressing): no per-lane 64-bit pointers kept across the loop
        rk[i] = *reinterpret_cast<const uint4*>(kb_ + (unsigned)(j * a.ldk + c * 8));
        if (HAS_POS) rk[2 + i] = *reinterpret_cast<const uint4*>(pkb_ + (unsigned)(j * a.ldpk + c * 8));
        rv[i] = *reinterpret_cast<const uint4*>(vb_ + (unsigned)(j * a.ldv + c * 8));
      }
    }
  };
  auto store_kv = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7;
      *reinterpret_cast<uint4*>(sKb(buf) + kx_off(r, c)) = rk[i];
      if (HAS_POS) *reinterpret_cast<uint4*>(sKb(buf) + kx_off(r, 8 + c)) = rk[2 + i];
      *reinterpret_cast<uint4*>(sVb(buf) + vx_off(r, c * 16)) = rv[i];
    }
  };

  f32x16 dq[NKS / 2];
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int c = 0; c < NKS / 2; ++c) dq[c][e] = 0.f;
  float pdp = 0.f;        // sum_j P_ij dP_ij (dP before the gain): d c_attn[h] = sum_{b,i} of it, exact for every c_attn

  if (nsched > 0) { load_kv(tile_of(0) * 64); store_kv(0); }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) consume_frag(qf[ks]);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) consume_frag(dof[ks]);
  __syncthreads();
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;

  for (int it = 0; it < nsched; ++it) {
    const int cur = it & 1;
    const int j0 = tile_of(it) * 64;
    if (it + 1 < nsched) load_kv(tile_of(it + 1) * 64);
    const bool tile_grid = j0 < Pk;
    // (waves whose 32 queries all lie past T only help staging the tiles)
    const bool skip = (qw >= a.T) || (a.causal && tile_grid && (j0 > qw + 31 || qw >= a.P));
    if (!skip) {
      // per-lane LDS offsets re-derived per tile from a re-materialised lane id (see the dK/dV kernel): K rows
      // (bK ^ (ks << 5)) + kb * 8192, V rows (bV ^ (ks << 5)) + kb * 4096, K^T ((bKt ^ (cb << 6)) + (kb * 32 + s2 * 16) * 256)
      // with the second row group at (+ 2048) ^ 32
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int half_i = ln >> 5, colT = ((ln >> 4) & 1) * 16 + (ln & 3) * 4, rT = 4 * half_i + ((ln & 15) >> 2);
      const int bK = kx_off(ln & 31, half_i), bV = vx_off(ln & 31, half_i * 16);
      const int bKt = kx_off(rT, colT >> 3) + (colT & 7) * 2;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int qw_u = __builtin_amdgcn_readfirstlane(qw);
        const bool wave_grid = qw_u + 31 < a.P;
        // 1: grid x grid; 2: straight-line body without a per-element bias -- no relative bias at all (cross attention),
        // or grid queries x text keys / text queries x grid keys, where the bias is one scalar per head (P % 32 == 0:
        // a wave lies on one side)
        const bool cbias = a.rel_mode && ((wave_grid && !tile_grid) || (qw_u >= a.P && tile_grid));
        const int fast = (a.rel_mode && tile_grid && wave_grid) ? 1 : ((cbias || (!a.rel_mode && !a.causal)) ? 2 : 0);
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = 0.f;
        auto sdp_mfma = [&]() {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 kf = lds_read_b128(sKb(cur) + ((bK ^ (ks << 5)) + kb * 8192));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            bf16x8 vfr = lds_read_b128(sVb(cur) + ((bV ^ (ks << 5)) + kb * 4096));
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, dof[ks], dp, 0, 0, 0);
          }
        };
        if (fast == 1 && row32) {
          // bias values of the key row at constant offsets from one table address seed the accumulator
          const float* tp = sTbl + (ciR + sGc[j0 + kb * 32] + 4 * half);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[rg * 4 + e] = tp[8 * rg + e];
          sdp_mfma();
        } else if (fast == 1 && rowseg) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float* tp = sTbl + (ciR + sGc[j0 + kb * 32 + 8 * rg] + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[rg * 4 + e] = tp[e];
          }
          sdp_mfma();
        } else {
          // (its own MFMA chain: see the forward kernel)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[e] = 0.f;
          sdp_mfma();
          if (cbias) {
            const float c0 = tile_grid ? relx1 : relx0;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] += c0;
          }
        }
        bf16x8 dsf[2];
        // the two hot block kinds get one straight-line body each, chosen ONCE per 32-key block: with the choice inside
        // the element loops every group of four elements is wrapped in its own wave-uniform branches (68 per tile)
        auto straight = [&](auto flags_tag) {
          constexpr int FLAGS = decltype(flags_tag)::value;       // 1: causal mask, 2: keys past S masked
          U128 ud[2];
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int dj = j0 + kb * 32 + 8 * rg + 4 * half - qi;     // masked (causal) iff key > query
            const int jl = a.S - (j0 + kb * 32 + 8 * rg + 4 * half);  // element e is a real key iff e < jl
            float dsv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float p = __builtin_amdgcn_exp2f(fmaf(s[rg * 4 + e], LOG2E, nlse_q));
              if (FLAGS & 1) p = (dj + e > 0) ? 0.f : p;
              if (FLAGS & 2) p = (e < jl) ? p : 0.f;
              dsv[e] = p * fmaf(gain, dp[rg * 4 + e], -del_q);
              pdp = fmaf(p, dp[rg * 4 + e], pdp);
            }
            ud[rg >> 1].w[(rg & 1) * 2] = pack2bf(dsv[0], dsv[1]);
            ud[rg >> 1].w[(rg & 1) * 2 + 1] = pack2bf(dsv[2], dsv[3]);
          }
          dsf[0] = ud[0].b; dsf[1] = ud[1].b;
        };
        if (fast == 1 && (row32 || rowseg)) {            // bias already in s (seeded accumulator)
          // only blocks crossing the diagonal hold masked elements (blocks entirely above it were skipped)
          if (a.causal && j0 + kb * 32 + 31 > qw_u) straight(std::integral_constant<int, 1>{});
          else straight(std::integral_constant<int, 0>{});
        } else if (fast == 2) {              // bias (if any) already in s
          if (j0 + kb * 32 + 32 > a.S) straight(std::integral_constant<int, 2>{});
          else straight(std::integral_constant<int, 0>{});
        } else
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          U128 ud;
#pragma unroll
          for (int rg2 = 0; rg2 < 2; ++rg2) {
            const int rg = s2 * 2 + rg2;
            const int jb = j0 + kb * 32 + 8 * rg + 4 * half;
            float dsv[4];
            if (fast == 1) {
              const int dj = jb - qi;
              float sv[4];
              if (row32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = s[rg * 4 + e];
              } else {
                const int4 cj = *reinterpret_cast<const int4*>(sGc + jb);
                const int cjs[4] = {cj.x, cj.y, cj.z, cj.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = s[rg * 4 + e] + sTbl[ciR + cjs[e]];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float p = __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, nlse_q));
                if (a.causal) p = (dj + e > 0) ? 0.f : p;
                dsv[e] = p * (gain * dp[rg * 4 + e] - del_q);
                pdp = fmaf(p, dp[rg * 4 + e], pdp);
              }
            } else {
              int4 cj = make_int4(0, 0, 0, 0);
              if (a.rel_mode && tile_grid) cj = *reinterpret_cast<const int4*>(sGc + jb);
              const int cjs[4] = {cj.x, cj.y, cj.z, cj.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int j = jb + e;
                float sv = s[rg * 4 + e];
                if (a.rel_mode) {
                  float bias;
                  if (tile_grid) bias = q_grid ? sTbl[ciR + cjs[e]] : relx1;
                  else bias = q_grid ? relx0 : ((j < a.S && qvalid) ? rel1d[ti - (j - a.P)] : 0.f);
                  sv += bias;
                }
                bool masked = j >= a.S;
                if (a.causal) {
                  if (tile_grid) masked |= (qi >= a.P) || (j > qi);
                  else masked |= (qi >= a.P) && (j > qi);
                }
                const float p = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(sv, LOG2E, nlse_q));
                dsv[e] = p * (gain * dp[rg * 4 + e] - del_q);
                pdp = fmaf(p, dp[rg * 4 + e], pdp);
              }
            }
            ud.w[rg2 * 2] = pack2bf(dsv[0], dsv[1]); ud.w[rg2 * 2 + 1] = pack2bf(dsv[2], dsv[3]);
          }
          dsf[s2] = ud.b;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int cb = 0; cb < NKS / 2; ++cb) {
            const int o0 = (bKt ^ (cb << 6)) + (kb * 32 + s2 * 16) * 256;
            U64 x, y;
            x.s = lds_read_tr(sKb(cur) + o0);
            y.s = lds_read_tr(sKb(cur) + ((o0 + 2048) ^ 32));
            U128 f; f.w[0] = x.w[0]; f.w[1] = x.w[1]; f.w[2] = y.w[0]; f.w[3] = y.w[1];
            dq[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, dsf[s2], dq[cb], 0, 0, 0);
          }
        }
      }
    }
    if (it + 1 < nsched) store_kv(cur ^ 1);
    __syncthreads();
  }

  if (a.dgain_rows) {
    const float tot = pdp + __shfl_xor(pdp, 32);
    if (qvalid && half == 0) a.dgain_rows[((long long)b * a.H + h) * a.T + qi] = tot;
  }
  if (qvalid) {
    bf16_t* dqp = a.dq + (long long)b * a.dq_bs + (long long)qi * a.lddq + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) store_tile_bf16(dqp + db * 32, dq[db], a.dq_scale, half, true);
    if constexpr (HAS_POS) {
      bf16_t* pp = a.dpq + ((long long)b * a.T + qi) * (a.H * 64) + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db) store_tile_bf16(pp + db * 32, dq[2 + db], a.dpq_scale, half, true);
    }
  }
}

template <bool HAS_POS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) { attn_bwd_dq_body<HAS_POS>(a, blockIdx.x); }

// (Measured and removed: ONE launch for the whole attention backward -- the dK/dV workgroups followed by the dQ workgroups in one
// grid.  Round 3: 447 us on the encoder shape against 189 + 125 us for the two kernels, 18.82 vs 17.89 ms per step: one kernel =
// one register allocation for both bodies, 75 spilled SGPRs and scratch in the loop.)

// delta[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d]   (8 lanes per (row, head))
__global__ void attn_delta_kernel(const bf16_t* o, const bf16_t* dO, float* delta, int B, int H, int T,
                                  long long o_bs, int ldo, long long do_bs, int lddo) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long grp = gid >> 3;
  const int sub = (int)(gid & 7);
  const long long total = (long long)B * T * H;
  float acc = 0.f;
  if (grp < total) {
    const int hh = (int)(grp % H);
    const long long bt = grp / H;
    const int t = (int)(bt % T), bb = (int)(bt / T);
    U128 x, y;
    x.v = *reinterpret_cast<const uint4*>(o + bb * o_bs + (long long)t * ldo + hh * 64 + sub * 8);
    y.v = *reinterpret_cast<const uint4*>(dO + bb * do_bs + (long long)t * lddo + hh * 64 + sub * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += bflo(x.w[e]) * bflo(y.w[e]) + bfhi(x.w[e]) * bfhi(y.w[e]);
  }
  acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
  if (grp < total && sub == 0) {
    const int hh = (int)(grp % H);
    const long long bt = grp / H;
    delta[((bt / T) * H + hh) * T + (bt % T)] = acc;
  }
}

}  // namespace

namespace {
// One launch for the reductions behind ifseg_attn_bwd's partial outputs; block ranges: [dpos_q | dpos_k | dgain | tables]
constexpr int SMALL_TAB_MAX = 2048;
struct ReduceArgs { ifseg_attn_reduce_args a; int nbq, nbk, nbt[3]; };
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(ReduceArgs r) {
  const ifseg_attn_reduce_args& a = r.a;
  int blk = blockIdx.x;
  const int tid = threadIdx.x;
  if (blk < r.nbq + r.nbk) {
    const bool isq = blk < r.nbq;
    if (!isq) blk -= r.nbq;
    const long long n = (long long)(isq ? a.T : a.S) * a.C;      // C % 8 == 0: eight columns (16 bytes of bf16) per thread
    const bf16_t* part = reinterpret_cast<const bf16_t*>(isq ? a.dpos_q_part : a.dpos_k_part);
    float* acc = isq ? a.dpos_q_acc : a.dpos_k_acc;
    const long long i = ((long long)blk * 256 + tid) * 8;
    if (i >= n) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.accumulate_pos) {
      const float4 a0 = *reinterpret_cast<const float4*>(acc + i), a1 = *reinterpret_cast<const float4*>(acc + i + 4);
      s[0] = a0.x; s[1] = a0.y; s[2] = a0.z; s[3] = a0.w; s[4] = a1.x; s[5] = a1.y; s[6] = a1.z; s[7] = a1.w;
    }
#pragma unroll 8
    for (int b = 0; b < a.B; ++b) {
      const uint4 v = *reinterpret_cast<const uint4*>(part + (long long)b * n + i);
      s[0] += bflo(v.x); s[1] += bfhi(v.x); s[2] += bflo(v.y); s[3] += bfhi(v.y);
      s[4] += bflo(v.z); s[5] += bfhi(v.z); s[6] += bflo(v.w); s[7] += bfhi(v.w);
    }
    *reinterpret_cast<float4*>(acc + i) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(acc + i + 4) = make_float4(s[4], s[5], s[6], s[7]);
    return;
  }
  blk -= r.nbq + r.nbk;
  if (blk < a.H) {          // d c_attn[h]
    if (!a.dgain) return;
    __shared__ float red[4];
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) {
      const float* d = a.delta + ((long long)b * a.H + blk) * a.T;
      for (int t = tid; t < a.T; t += 256) s += d[t];
    }
    s = warp_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) reinterpret_cast<bf16_t*>(a.dgain)[blk] = f2bf(((red[0] + red[1]) + (red[2] + red[3])) / a.gain[blk]);
    return;
  }
  blk -= a.H;
  __shared__ float tred[8][33];
  __shared__ float tsmall[SMALL_TAB_MAX];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t >= a.ntab) return;
    if (blk < r.nbt[t] && a.tab_n[t] <= SMALL_TAB_MAX) {
      // small table (token offsets, bos row / column): one block per head sums the partials of every entry into LDS,
      // then thread r adds the entries of bucket r in entry order -- several entries may share a bucket (log-spaced
      // buckets beyond +-128), and a fixed order keeps the sum bit-reproducible (no atomics)
      const int n = a.tab_n[t], h = blk;
      if (a.nparts <= 8) {
        // few partial tables (the batch-inner path hands over 4): a thread per entry, one pass -- the grouped passes below
        // cost two barriers per 32 entries (14 passes for the 429 offsets of a 215-token prompt).  Same order of additions.
        for (int j = tid; j < n; j += 256) {
          const float* p = a.tab_part[t] + (long long)h * a.nparts * n + j;
          float sum = 0.f;
          for (int q = 0; q < a.nparts; ++q) sum += p[(long long)q * n];
          tsmall[j] = sum;
        }
        __syncthreads();
      } else
      for (int j0 = 0; j0 < n; j0 += 32) {          // 32 entries x 8 groups of partials per pass
        const int j = j0 + (tid & 31), g = tid >> 5;
        float sum = 0.f;
        if (j < n) {
          const float* p = a.tab_part[t] + (long long)h * a.nparts * n + j;
#pragma unroll 4
          for (int q = g; q < a.nparts; q += 8) sum += p[(long long)q * n];
        }
        tred[g][tid & 31] = sum;
        __syncthreads();
        if (g == 0 && j < n) {
          float tot = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) tot += tred[q][tid];
          tsmall[j] = tot;
        }
        __syncthreads();
      }
      // (the entry -> bucket map staged in LDS: every thread walks all n entries, and n dependent-latency global loads per
      // thread -- 429 for a 215-token prompt -- were most of this block's time)
      __shared__ int sidx[SMALL_TAB_MAX];
      for (int j = tid; j < n; j += 256) sidx[j] = a.tab_idx[t][j];
      __syncthreads();
      for (int rb = tid; rb < a.tab_nbucket[t]; rb += 256) {
        float tot = 0.f;
        bool any = false;
        for (int j = 0; j < n; ++j)
          if (sidx[j] == rb) { tot += tsmall[j]; any = true; }
        if (any) a.tab_acc[t][(long long)rb * a.H + h] += tot;
      }
      return;
    }
    if (blk < r.nbt[t]) {
      // 32 table entries x 8 groups of partials per block (the partials of one entry are nparts * n floats apart)
      const int n = a.tab_n[t], nchunk = (n + 31) / 32;
      const int h = blk / nchunk, j = (blk - h * nchunk) * 32 + (tid & 31), g = tid >> 5;
      float sum = 0.f;
      if (j < n) {
        const float* p = a.tab_part[t] + (long long)h * a.nparts * n + j;
#pragma unroll 4
        for (int q = g; q < a.nparts; q += 8) sum += p[(long long)q * n];
      }
      tred[g][tid & 31] = sum;
      __syncthreads();
      if (g == 0 && j < n) {
        const int bucket = a.tab_idx[t][j];
        if (bucket >= 0) {
          float tot = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) tot += tred[q][tid];
          atomicAdd(&a.tab_acc[t][(long long)bucket * a.H + h], tot);      // (large tables map entries to buckets one-to-one)
        }
      }
      return;
    }
    blk -= r.nbt[t];
  }
}
}  // namespace

extern "C" int ifseg_attn_bwd_reduce(const ifseg_attn_reduce_args* x, void* stream) {
  (void)hipGetLastError();
  if (!x || x->B <= 0 || x->H <= 0 || x->T <= 0 || x->S <= 0 || (x->C & 7) || x->ntab < 0 || x->ntab > 3) return IFSEG_ERR_BAD_ARG;
  ReduceArgs r{};
  r.a = *x;
  r.nbq = x->dpos_q_part ? (int)(((long long)x->T * x->C / 8 + 255) / 256) : 0;
  r.nbk = x->dpos_k_part ? (int)(((long long)x->S * x->C / 8 + 255) / 256) : 0;
  long long total = (long long)r.nbq + r.nbk + x->H;
  for (int t = 0; t < x->ntab; ++t) {
    if (!x->tab_part[t] || !x->tab_idx[t] || !x->tab_acc[t] || x->tab_n[t] <= 0 || x->nparts <= 0) return IFSEG_ERR_BAD_ARG;
    if (x->tab_nbucket[t] <= 0) return IFSEG_ERR_BAD_ARG;
    r.nbt[t] = x->tab_n[t] <= SMALL_TAB_MAX ? x->H : x->H * ((x->tab_n[t] + 31) / 32);
    total += r.nbt[t];
  }
  if ((x->dgain && (!x->delta || !x->gain)) || total >= (1ll << 31)) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, r);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

static int attn_check(const AttnArgs& a) {
  if (a.rel_mode || a.causal) {
    if (a.P % 64 || a.P > a.T || a.P > a.S) return IFSEG_ERR_BAD_SHAPE;
  }
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return IFSEG_ERR_BAD_SHAPE;
  // rows are read and written 16 bytes at a time: bases 16-byte aligned, batch strides multiples of 8 elements
  if (((size_t)a.q | (size_t)a.k | (size_t)a.v | (size_t)a.o | (size_t)a.dO | (size_t)a.dq | (size_t)a.dk | (size_t)a.dv) & 15)
    return IFSEG_ERR_BAD_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs | a.do_bs | a.dq_bs | a.dk_bs | a.dv_bs) & 7) return IFSEG_ERR_BAD_SHAPE;
  return 0;
}

extern "C" int ifseg_attn_fwd(const void* q, const void* k, const void* v, const void* pos_q, const void* pos_k,
                              void* out, float* lse, int B, int H, int T, int S, int ldq, int ldk, int ldv,
                              int ldo, int ldpq, int ldpk, long long q_bs, long long k_bs, long long v_bs,
                              long long o_bs, int rel_mode, int P, const int* gcode, int code_bias, int n2d,
                              const float* rel2d, const float* rel1d, const float* relx, int causal,
                              const float* dense_bias, const void* gain, int grid_w, int dense_ld, void* stream) {
  (void)hipGetLastError();
  AttnArgs a{};
  a.grid_w = grid_w;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.pq = (const bf16_t*)pos_q; a.pk = (const bf16_t*)pos_k; a.o = (bf16_t*)out; a.lse = lse;
  a.B = B; a.H = H; a.T = T; a.S = S; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.ldpq = ldpq; a.ldpk = ldpk; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.rel_mode = rel_mode; a.P = P; a.gcode = gcode; a.code_bias = code_bias; a.n2d = rel_mode ? n2d : 0;
  a.Lt = T - P; a.rel2d = rel2d; a.rel1d = rel1d; a.relx = relx; a.causal = causal; a.dense = dense_bias; a.gain = (const float*)gain;
  a.dense_ld = dense_bias ? (dense_ld > 0 ? dense_ld : S) : 0;
  if (dense_bias && a.dense_ld < S) return IFSEG_ERR_BAD_ARG;
#ifdef IFSEG_EXP_NOBIAS_FWD
  pos_q = nullptr; a.pq = nullptr; a.pk = nullptr; a.rel_mode = 0; a.n2d = 0; rel_mode = 0;   // timing bound only
#endif
  if (!rel_mode && !causal) a.P = S;
  int rc = attn_check(a);
  if (rc) return rc;
  const int nq = (T + 127) / 128;
  size_t lds = 2 * (KT_BYTES + VT_BYTES) + (rel_mode ? (((size_t)a.n2d + 3) & ~3) * 4 + (size_t)P * 4 : 0);
  if (lds > 160 * 1024) return IFSEG_ERR_BAD_SHAPE;
  dim3 grid(nq * H * B), block(256);
  hipStream_t s = (hipStream_t)stream;
  // algorithmic flops of the reference: QK^T + PV at head dim 64 (dense), 4*T*S*64 per (b,h)
  ifseg_prof_begin(IFSEG_K_ATTN_FWD, s, 4.0 * 64 * (double)T * S * B * H, 0);
  if (pos_q) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, block, lds, s, a);
  } else {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, block, lds, s, a);
  }
  ifseg_prof_end(IFSEG_K_ATTN_FWD, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_attn_bwd(const ifseg_attn_bwd_args* x, void* stream) {
  (void)hipGetLastError();
  AttnArgs a{};
  a.q = (const bf16_t*)x->q; a.k = (const bf16_t*)x->k; a.v = (const bf16_t*)x->v;
  a.pq = (const bf16_t*)x->pos_q; a.pk = (const bf16_t*)x->pos_k; a.lse = (float*)x->lse;
  a.B = x->B; a.H = x->H; a.T = x->T; a.S = x->S;
  a.ldq = x->ldq; a.ldk = x->ldk; a.ldv = x->ldv; a.ldpq = x->ldpq; a.ldpk = x->ldpk;
  a.q_bs = x->q_bs; a.k_bs = x->k_bs; a.v_bs = x->v_bs;
  a.rel_mode = x->rel_mode; a.P = x->P; a.gcode = x->gcode; a.code_bias = x->code_bias;
  a.n2d = x->rel_mode ? x->n2d : 0; a.Lt = x->T - x->P;
  a.rel2d = x->rel2d; a.rel1d = x->rel1d; a.relx = x->relx; a.causal = x->causal;
  a.dO = (const bf16_t*)x->dout; a.do_bs = x->do_bs; a.lddo = x->lddo; a.delta = x->delta;
  a.dq = (bf16_t*)x->dq; a.dk = (bf16_t*)x->dk; a.dv = (bf16_t*)x->dv;
  a.dq_bs = x->dq_bs; a.dk_bs = x->dk_bs; a.dv_bs = x->dv_bs; a.lddq = x->lddq; a.lddk = x->lddk; a.lddv = x->lddv;
  a.dpq = (bf16_t*)x->dpos_q_part; a.dpk = (bf16_t*)x->dpos_k_part;
  a.drel2d_part = x->drel2d_part; a.drel1d_part = x->drel1d_part; a.drelx_part = x->drelx_part;
  a.nparts = x->nparts; a.gain = (const float*)x->gain; a.dq_scale = x->dq_scale; a.dpq_scale = x->dpq_scale;
  a.grid_w = x->grid_w;
  a.dgain_rows = x->dgain_rows;
#ifdef IFSEG_EXP_NOBIAS_BWD
  // timing bound only (tools/variant.py): the backward with every per-batch bias term compiled out -- no abs-pos columns,
  // no rel-pos seeds, no table-gradient bins (results are wrong)
  a.pq = nullptr; a.pk = nullptr; a.rel_mode = 0; a.n2d = 0;
#endif
  if (!a.rel_mode && !a.causal) a.P = a.S;
  int rc = attn_check(a);
  if (rc) return rc;
  if ((x->lddo | x->lddq | x->lddk | x->lddv | x->ldout) & 7) return IFSEG_ERR_BAD_SHAPE;
  {   // the dK/dV kernel addresses a query row by a 32-bit byte offset from the batch element's base
    const long long ldmax = a.ldq > a.lddo ? (a.ldq > a.ldpq ? a.ldq : a.ldpq) : (a.lddo > a.ldpq ? a.lddo : a.ldpq);
    if ((long long)a.T * ldmax * 2 >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  }
  const int nkt = (a.S + 127) / 128, nq = (a.T + 127) / 128;
  if (a.rel_mode && a.nparts != a.B * nkt) return IFSEG_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int ph = x->phases ? x->phases : (IFSEG_ATTN_BWD_DELTA | IFSEG_ATTN_BWD_DKV | IFSEG_ATTN_BWD_DQ);
  if (ph & IFSEG_ATTN_BWD_DELTA) {  // delta = rowsum(dO * O)
    const long long threads = (long long)a.B * a.T * a.H * 8;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                       (const bf16_t*)x->out, a.dO, x->delta, a.B, a.H, a.T, x->out_bs, x->ldout, a.do_bs, a.lddo);
  }
  const size_t n2dp = ((size_t)a.n2d + 3) & ~(size_t)3;
  const size_t n1dp = a.rel_mode ? (((size_t)(2 * a.Lt - 1) + 3) & ~(size_t)3) : 0;
  const bool rowseg_h = a.rel_mode && a.grid_w != 32 && a.grid_w >= 32 && (a.grid_w & 7) == 0;   // as in the kernel
  auto lds_kv_of = [&](int nw) {
    return (size_t)(KT_BYTES + VT_BYTES + 512) + (nw / 2) * VT_BYTES +
           (a.rel_mode ? (2 * n2dp + nw * n1dp + 2 * nw) * 4 + (((size_t)a.P + 3) & ~(size_t)3) * 4 +
                             (rowseg_h ? (size_t)2 * nw * (4 + 3 * 64) * 4 : 0) : 0);      // exchange area: `rowseg` grids only
  };
  // 8-wave workgroups (one copy of the tables for twice the waves) when a 4-wave workgroup takes more than half a CU's LDS
  const bool dkv8 = a.rel_mode && lds_kv_of(4) > 80 * 1024 && lds_kv_of(8) <= 160 * 1024;
  const size_t lds_kv = lds_kv_of(dkv8 ? 8 : 4);
  const size_t lds_q = 2 * (KT_BYTES + VT_BYTES) + (a.rel_mode ? n2dp * 4 + (size_t)a.P * 4 : 0);
  if (lds_kv > 160 * 1024 || lds_q > 160 * 1024) return IFSEG_ERR_BAD_SHAPE;
  const bool do_kv = ph & IFSEG_ATTN_BWD_DKV, do_q = ph & IFSEG_ATTN_BWD_DQ;
#ifdef IFSEG_EXP_NOBIAS_BWD
  const bool has_pos = false;
#else
  const bool has_pos = x->pos_q != nullptr;
#endif
  if (has_pos) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (do_kv) {
      ifseg_prof_begin(IFSEG_K_ATTN_DKV, s, 6.0 * 64 * (double)a.T * a.S * a.B * a.H, 0);   // dV, dP, dK of the reference
      if (dkv8) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        hipLaunchKernelGGL(attn_bwd_dkv8_kernel<true>, dim3(((a.S + 255) / 256) * a.H * a.B), dim3(512), lds_kv, s, a);
      } else {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3(nkt * a.H * a.B), dim3(256), lds_kv, s, a);
      }
      ifseg_prof_end(IFSEG_K_ATTN_DKV, s);
    }
    if (do_q) {
      ifseg_prof_begin(IFSEG_K_ATTN_DQ, s, 2.0 * 64 * (double)a.T * a.S * a.B * a.H, 0);    // dQ of the reference
      hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(nq * a.H * a.B), dim3(256), lds_q, s, a);
      ifseg_prof_end(IFSEG_K_ATTN_DQ, s);
    }
  } else {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (do_kv) {
      if (dkv8) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        hipLaunchKernelGGL(attn_bwd_dkv8_kernel<false>, dim3(((a.S + 255) / 256) * a.H * a.B), dim3(512), lds_kv, s, a);
      } else {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3(nkt * a.H * a.B), dim3(256), lds_kv, s, a);
      }
    }
    if (do_q) hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(nq * a.H * a.B), dim3(256), lds_q, s, a);
  }
  IFSEG_CHECK_LAUNCH();
  return 0;
}

int ifseg_exp_attention() {
#if defined(IFSEG_EXP_NOBIAS_FWD) || defined(IFSEG_EXP_NOBIAS_BWD)
  return 1;
#else
  return 0;
#endif
}
