// Attention backward with the batch as the INNER dimension of a workgroup ("bi"), gfx950.
//
// Reference semantics (unify_multihead_attention.py:346,459-512 under autograd, with the bias the reference builds ONCE per
// layer and broadcasts over the batch: encoder_module.py:757-771,790-809, decoder_module.py:553-558,603-627, expand at
// encoder_module.py:317,791):
//     S_b = (q_b*scaling) k_b^T + Bias,   Bias[h] = abs_pos + rel_pos (+ causal / key-padding -inf): batch invariant
//     P_b = softmax_fp32(S_b);  O_b = gain[h] P_b v_b
//     dV_b = gain P_b^T dO_b;  dS_b = P_b o (gain dO_b v_b^T - delta_b);  dQ_b = dS_b k_b;  dK_b = dS_b^T q_b
//     dBias = sum_b dS_b      -> d abs-pos operands, d rel-pos tables (ifseg_attn_dbias_grads)
//
// MI355X formulation.  The bias is a dense bf16 operand D[h] ([Tp][Sp], padding and masked entries -inf; round 6: bf16 -- what the
// reference itself holds under --fp16, unify_multihead_attention.py:464, encoder_module.py:757-771 -- half the bytes of the operand
// in HBM and in every stage) built once per layer and step by
// ifseg_attn_dense_bias from parameters only (side stream, start of the step).  A workgroup of 8 waves owns
// (head, 64 stationary rows = 2 blocks of 32, 4 batch elements): wave = (row block, batch element).  The 4 batch waves of a
// row block read the SAME 32 x 32 bias tile from LDS (one LDS-DMA per tile instead of four regenerations by MFMA + table
// look-ups), so the hot loop has one uniform body for every bias kind (grid of any width, text, bos, causal, cross):
// the tile seeds the score accumulator, masked entries are -inf inside it.  In the dQ kernel the four waves leave their
// fp32 dS tiles in LDS and each sums a quarter of the four: sum_b dS leaves the kernel once per tile (bf16, one slab per
// group of 4 batch elements) -- no per-batch bias-gradient work, no abs-pos columns in the contractions (head dim 64
// everywhere), no table-gradient bins in the loop.  Deterministic: fixed summation order, no atomics.
//
//   dQ kernel   (lane = query):  S^T = K Q^T (A = K rows from LDS, B = q regs), dP^T = V dO^T, dS^T, dQ^T += K^T dS^T
//   dK/dV kernel (lane = key):   S = Q K^T (A = Q rows from LDS, B = k regs), dP = dO V^T (B = -gain V regs, accumulator
//                                seeded with delta), dV^T += dO^T P, dK^T += Q^T dS
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;

struct BiArgs {
  const bf16_t *q, *k, *v, *dO;
  const float *lse, *delta, *gain;
  const bf16_t* D;
  float* dgain_rows;
  bf16_t *dq, *dk, *dv, *dbias;
  bf16_t* out; float* lse_out; long long o_bs; int ldo;      // forward
  int B, H, T, S, Sp, Tp;
  long long q_bs, k_bs, v_bs, do_bs, dq_bs, dk_bs, dv_bs, dbias_gs;
  int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  int causal, P;
  float dq_scale;
  const int* kv_len;          // optional [B]: keys at or beyond kv_len[b] are padding (masked for batch element b)
  float drop_p;               // attention dropout (unify_multihead_attention.py:498): probability, 0 = off
  unsigned long long drop_seed; const unsigned long long* drop_seed_add;
};

// Attention dropout: P_drop = P o keep / (1 - p) between the softmax and P V (unify_multihead_attention.py:498-512).  keep is a
// counter-based hash of (seed, (b, h, query), key) -- the three kernels regenerate the same mask whichever of query / key is
// their lane, nothing is stored; the RNG stream necessarily differs from torch's.  The softmax denominator (lse) is that of
// the undropped P; delta = dO . O already contains the mask (rowsum(P_drop o dP_drop)), so
//     dS = P o (keep / (1 - p) * gain dO v^T - delta),   dV = gain P_drop^T dO.
struct AttnDrop { unsigned lo, hi, thr; float inv; };
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ AttnDrop attn_drop_setup(float p, unsigned long long seed, const unsigned long long* seed_add) {
  const unsigned long long sd = seed + (seed_add ? *seed_add : 0ull);
  AttnDrop d;
  d.lo = (unsigned)sd; d.hi = (unsigned)(sd >> 32);
  d.thr = (unsigned)(p * 16777216.f);
  d.inv = 1.f / (1.f - p);
  return d;
}
__device__ __forceinline__ unsigned attn_row_key(const AttnDrop& d, unsigned rowid) { return mix32(d.lo ^ rowid) ^ d.hi; }
// keep / (1 - p) of score (row, key j)
__device__ __forceinline__ float attn_keep(const AttnDrop& d, unsigned rk, int j) {
  return (mix32(rk + (unsigned)j * 0x9E3779B9u) >> 8) >= d.thr ? d.inv : 0.f;
}

// 32 x (128-byte row) tile image read BOTH row-wise (ds_read_b128: 16 rows x one 16-byte chunk per lane group) and
// transposed (ds_read_b64_tr_b16: 4 consecutive rows x one 64-byte granule): chunk index XOR-ed with a bit-rotated row id
// (two rows per 256-byte bank line).  Used for the bf16 operand tiles [32][64] and the fp32 bias tiles [32][32].
__device__ __forceinline__ int vx_off(int r, int colbyte) {
  const int u = r >> 1, hsw = ((u & 1) << 2) | ((u >> 1) & 3);
  return r * 128 + ((((colbyte >> 4) ^ hsw) & 7) << 4) + (colbyte & 15);
}
__device__ __forceinline__ int vx_swz(int r) { const int u = r >> 1; return ((u & 1) << 2) | ((u >> 1) & 3); }

__device__ __forceinline__ void consume_frag(const bf16x8& f) {
  typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
  asm volatile("" :: "v"(__builtin_bit_cast(u32x4_t, f)));
}

// Key padding (unify_multihead_attention.py:477-489: attn_weights.masked_fill(key_padding_mask, -inf); the padding mask of a
// batch element is a suffix of its key sequence, encoder_module.py:730-752): score seeds of keys at or beyond the element's
// valid count become -inf (the MFMA adds the products onto the seed: -inf stays).  Layout of a tile whose lane is the QUERY:
// element e <-> key j0 + (e & 3) + 8 (e >> 2) + 4 half.
__device__ __forceinline__ void mask_padded_keys(f32x16& s, int j0, int half, int kl) {
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (j0 + (e & 3) + 8 * (e >> 2) + 4 * half >= kl) s[e] = NEG_INF;
}

__device__ __forceinline__ uint4 scale_bf16x8(uint4 v, float f) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    w[i] = pack2bf(__uint_as_float(w[i] << 16) * f, __uint_as_float(w[i] & 0xffff0000u) * f);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// Epilogue of a 32 x 32 accumulator tile whose row is the lane: element r of lane (half, x) is column
// (r&3) + 8*(r>>2) + 4*half.  The two lanes of a row exchange one 8-byte run each so that every lane stores 16 bytes.
__device__ __forceinline__ void store_tile_bf16(bf16_t* rowp, const f32x16& acc, float scale, int half, bool valid) {
#pragma unroll
  for (int rgp = 0; rgp < 2; ++rgp) {
    const int e0 = rgp * 8, e1 = rgp * 8 + 4;
    const unsigned x0 = pack2bf(acc[e0] * scale, acc[e0 + 1] * scale), x1 = pack2bf(acc[e0 + 2] * scale, acc[e0 + 3] * scale);
    const unsigned y0 = pack2bf(acc[e1] * scale, acc[e1 + 1] * scale), y1 = pack2bf(acc[e1 + 2] * scale, acc[e1 + 3] * scale);
    const auto p0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
    const auto p1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    if (valid) *reinterpret_cast<uint4*>(rowp + 16 * rgp + 8 * half) = make_uint4(p0[0], p1[0], p0[1], p1[1]);
  }
}

// LDS image of one stage (both kernels): four batch elements' two operand tiles, two bias tiles, the dK/dV kernel's
// row statistics
constexpr int ST_A = 0;                 // dQ: K[4][32][64]   dK/dV: Q[4][32][64]       (bf16, 4 KiB each)
constexpr int ST_B = 16384;             // dQ: V[4][32][64]   dK/dV: dO[4][32][64]
constexpr int ST_D = 32768;             // bias tiles [2][32][32] bf16 (2 KiB each): 64-byte rows, 16-byte chunk ^= (row >> 2) & 3
constexpr int ST_L = 36864;             // dK/dV: [4][lse 32 | delta 32] fp32
constexpr int STG_DQ = 36864, STG_DKV = 37888;
constexpr int SLOT = 4096;              // dQ kernel: one wave's fp32 dS tile [32][32], chunk XOR (row & 7)
constexpr int LDS_DQ = 2 * STG_DQ + 2 * 8 * SLOT;      // 139264
constexpr int LDS_DKV = 2 * STG_DKV;                    // 75776

// bias tile image [32 rows][32 keys] bf16: row r at r * 64 bytes, its four 16-byte chunks XOR-ed with (r >> 2) & 3 (the LDS-DMA
// granule is 16 bytes: the swizzle is applied on the source address).  Row-wise readers (lane = query: 8 bytes = the lane's four
// consecutive keys of a register group) see a 2-way conflict = the cycles of the fp32 tile's 16-byte reads; the dK/dV kernel reads
// four consecutive QUERIES of one key per lane with the hardware transpose (ds_read_b64_tr_b16): four rows x 64 bytes per
// 32 lanes, every bank once.
__device__ __forceinline__ int db_swz(int r) { return (r >> 2) & 3; }
__device__ __forceinline__ int db_off(int r, int colbyte) { return r * 64 + ((((colbyte >> 4) ^ db_swz(r)) & 3) << 4) + (colbyte & 15); }
// one LDS-DMA piece of a bias tile: rows 16 p .. 16 p + 15 (lane l: row 16 p + (l >> 2), position l & 3)
__device__ __forceinline__ void db_stage(const bf16_t* dbase, int p, int row0, int rmax, int Sp, int col0, int lane, unsigned dst) {
  const int row = p * 16 + (lane >> 2);
  const int c = (lane & 3) ^ db_swz(row);
  const int ir = min(row0 + row, rmax);
  lds_dma16_gs(dbase, (ir * Sp + col0 + c * 8) * 2, dst + p * 1024);
}

// ---- staging through BUFFER loads (round 6): the part of a piece's source address that changes from block to block -- the
// streamed side's first row -- is wave-uniform, so it rides in the instruction's SCALAR offset; the per-lane part (row inside
// the piece, swizzled chunk) is the same for every block, and rows past the end of the tensor are out of the descriptor's
// range (zeros: their scores are -inf from the bias padding, their P is 0).  With global_load_lds every piece of every block
// recomputed row, clamp, swizzle, multiply and add per lane: ~50 of a block's ~160 VALU instructions in kernels whose VALU
// work is 3x their MFMA cycles.
typedef int bi_v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bi_v4i32 bi_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  bi_v4i32 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16_bs(bi_v4i32 rs, unsigned lds_base, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
// per-lane byte offsets of an operand tile's pieces [32 rows][64 bf16]: piece p = rows 8 p + (lane >> 3), chunk (lane & 7) ^
// vx_swz(row); vx_swz(8 p + r) = vx_swz(r) ^ (2 (p & 1)), so even pieces share `even`, odd pieces `odd`; + the scalar (first row + 8 p) * ld * 2
__device__ __forceinline__ void tile_voff(int lane, int ld, unsigned& even, unsigned& odd) {
  const int r8 = lane >> 3, c0 = (lane & 7) ^ vx_swz(r8);
  const unsigned rb = (unsigned)(r8 * ld) * 2u;
  even = rb + ((unsigned)c0 << 4);
  odd = rb + ((unsigned)(c0 ^ 2) << 4);
}
// per-lane byte offset of piece p of a bias tile whose first row is row0 (rows clamped to rmax) and whose columns start at col0
__device__ __forceinline__ unsigned db_voff(int p, int row0, int rmax, int Sp, int col0, int lane) {
  const int row = p * 16 + (lane >> 2);
  const int c = (lane & 3) ^ db_swz(row);
  return (unsigned)((min(row0 + row, rmax) * Sp + col0 + c * 8) * 2);
}
// four bf16 seeds of register group rg (8 bytes, requested with the block's other LDS reads) -> fp32, right before the MFMAs:
// between the request and the first MFMA a block holds 8 registers of raw seeds instead of 16 of fp32 ones
__device__ __forceinline__ void db_expand(const uint2& w, f32x16& s, int rg) {
  s[rg * 4] = bflo(w.x); s[rg * 4 + 1] = bfhi(w.x); s[rg * 4 + 2] = bflo(w.y); s[rg * 4 + 3] = bfhi(w.y);
}

// Block schedule of the streamed side under the causal mask ("tail-first" order: a grid row i sees grid columns j <= i and
// every tail column; a tail row sees tail columns j <= i only).  The dense bias already holds -inf for every masked
// entry; the schedule only skips 32-blocks that are masked entirely.  first..g_end-1 are grid blocks, then tail blocks.
struct Sched {
  int g_begin, g_end, t_begin, n;
  __device__ __forceinline__ int block(int it) const { return it < g_end - g_begin ? g_begin + it : t_begin + (it - (g_end - g_begin)); }
};

// Loop structure of both kernels: one s_barrier per 32-row block of the streamed side; the stage of block n+1 is in flight
// (LDS-DMA, issued right after the barrier) while block n is computed; every LDS read of a block is requested before its first
// MFMA.  What was measured on the way (encoder shape, B = 8, dK/dV kernel alone; DESIGN.md "Round 4" (1)):
//   v1  lock-step waves, reads next to their MFMAs ................................ 122.6 us
//   v2  the two wave groups of a SIMD phase-shifted inside the barrier interval ..... 121.6 us  (no gain: dropped)
//   v3  all LDS reads of a block up front ......................................... 106.2 us  (kept)
//   v4  + gradient MFMAs of block n-1 under the row reads of block n .................. 110.7 us  (dropped)
//   v5  + staging two blocks ahead, ring of three stages ............................ 112.2 us  (dropped)
// Ablation of v2 (profiles/round4_attn_bi_ablation.txt; the variants were -D switches of that version of this file, since removed): without the S/dP MFMAs 109.8, without exp 115.8, without the gradient MFMAs 96.0,
// without any MFMA 88.2, without any MFMA and without staging 67.8, staging + barriers alone 67.3 us: two waves per SIMD do
// not hide the block's dependent LDS round trips, and no single pipe is the bound.

// ---------------------------------------------------------------------------------------------- forward
// O_b = gain softmax(q_b k_b^T + D) v_b for four batch elements per workgroup (wave = (32-query block, batch element)): the
// 32 x 32 bias tile is staged once for the four, no abs-pos columns in the contraction, no table look-ups, one path for every
// bias kind (unify_multihead_attention.py:459-512; the bias of encoder_module.py:757-809 / decoder_module.py:553-627 is the
// dense operand D).  Flash-style online softmax in the exp2 domain with a lazily raised reference maximum (as csrc/attention.hip);
// computed swapped (S^T = K Q^T, O^T = V^T P^T) so a query is a lane.
constexpr float LAZY_MAX_SLACK = 8.f;
template <bool DROP>
__device__ __forceinline__ void attn_bi_fwd_body(const BiArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wv >> 2, bl = wv & 3;
  const int nqt = (a.T + 63) >> 6, nbg = (a.B + 3) >> 2;
  const int bid = xcd_remap(blockIdx.x, nqt * a.H * nbg);
  const int bg = bid % nbg, h = (bid / nbg) / nqt;
  int qt = (bid / nbg) % nqt;
  if (a.causal) qt = nqt - 1 - qt;
  const int q0 = qt * 64;
  const int b = bg * 4 + bl;
  const bool bact = b < a.B;
  const int bc = bact ? b : a.B - 1;
  const int qi = q0 + qb * 32 + (lane & 31);
  const bool qvalid = qi < a.T;
  const int qrow = qvalid ? qi : a.T - 1;
  bf16x8 qf[4];
  {
    const bf16_t* qp = a.q + (long long)bc * a.q_bs + (long long)qrow * a.ldq + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(qp + ks * 16); qf[ks] = u.b; }
  }
  Sched sc;
  {
    const int nkb = a.Sp >> 5;
    if (a.causal) {
      const int pb = a.P >> 5;
      sc.g_begin = 0;
      sc.g_end = q0 < a.P ? min(pb, (min(q0 + 63, a.P - 1) >> 5) + 1) : 0;
      sc.t_begin = pb;
      sc.n = sc.g_end + (nkb - pb);
    } else {
      sc.g_begin = 0; sc.g_end = nkb; sc.t_begin = nkb; sc.n = nkb;
    }
  }
  const bf16_t* kb_ = a.k + (long long)bc * a.k_bs + h * 64;
  const bf16_t* vb_ = a.v + (long long)bc * a.v_bs + h * 64;
  const bf16_t* db_ = a.D + (long long)h * a.Tp * a.Sp;
  const unsigned lds0 = lds_addr(smem);
  // this wave stages V (row block 0) or K (row block 1) of its batch element: rows 0 .. S-1, 64 columns of head h
  const int ldkv = qb ? a.ldk : a.ldv;
  const bi_v4i32 rsKV = bi_rsrc(qb ? kb_ : vb_, (unsigned)(((long long)(a.S - 1) * ldkv + 64) * 2));
  const bi_v4i32 rsD = bi_rsrc(db_, (unsigned)((long long)a.Tp * a.Sp * 2));
  unsigned ve, vo;                                            // (three loop-invariant per-lane offsets)
  tile_voff(lane, ldkv, ve, vo);
  const unsigned vD = db_voff(bl & 1, q0 + qb * 32, a.T - 1, a.Sp, 0, lane);
  const int kl = a.kv_len ? __builtin_amdgcn_readfirstlane(a.kv_len[bc]) : 0x7fffffff;
  AttnDrop dr{};
  unsigned rk = 0;
  if (DROP) { dr = attn_drop_setup(a.drop_p, a.drop_seed, a.drop_seed_add); rk = attn_row_key(dr, ((unsigned)bc * a.H + h) * a.T + qi); }
  auto issue = [&](int it, int st) {
    const int j0 = sc.block(it) * 32;
    const unsigned base = lds0 + st * STG_DQ;
    const unsigned dst = base + (qb ? ST_A : ST_B) + bl * 4096;
    const unsigned s0 = (unsigned)(j0 * ldkv) * 2u, sp = (unsigned)ldkv * 16u;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) lds_dma16_bs(rsKV, dst + piece * 1024, (piece & 1) ? vo : ve, s0 + piece * sp);
    if (bl < 2) lds_dma16_bs(rsD, base + ST_D + qb * 2048 + bl * 1024, vD, (unsigned)j0 * 2u);
  };
  f32x16 oacc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
  float m_run = NEG_INF, l_run = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) consume_frag(qf[ks]);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  int oR[4], oT[2][2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) oR[ks] = vx_off(lane & 31, half * 16) ^ (ks << 5);
  const int oD = db_off(lane & 31, half * 8);            // register group rg: ^ (rg << 4)
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int colb = (db * 32 + g16 * 16 + (i16 & 3) * 4) * 2, r0 = 16 * s2 + 4 * half + (i16 >> 2);
      oT[s2][db][0] = vx_off(r0, colb); oT[s2][db][1] = vx_off(r0 + 8, colb);
    }

  if (sc.n > 0) issue(0, 0);
  for (int it = 0; it < sc.n; ++it) {
    lds_dma_wait();
    __syncthreads();
    const unsigned char* stg = smem + (it & 1) * STG_DQ;
    const unsigned char* sK = stg + ST_A + bl * 4096;
    const unsigned char* sV = stg + ST_B + bl * 4096;
    const unsigned char* sD = stg + ST_D + qb * 2048;
    f32x16 s;
    bf16x8 kf[4];
    U128 vt[2][2];
    uint2 wD[4];
    {
      int ll = lane;
      asm volatile("" : "+v"(ll));                       // (re-derived per block: four hoisted offsets cost the kernel its second workgroup per CU)
      const int oDl = db_off(ll & 31, (ll >> 5) * 8);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) wD[rg] = *reinterpret_cast<const uint2*>(sD + (oDl ^ (rg << 4)));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = lds_read_b128(sK + oR[ks]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        U64 x, y;
        x.s = lds_read_tr(sV + oT[s2][db][0]);
        y.s = lds_read_tr(sV + oT[s2][db][1]);
        vt[s2][db].w[0] = x.w[0]; vt[s2][db].w[1] = x.w[1]; vt[s2][db].w[2] = y.w[0]; vt[s2][db].w[3] = y.w[1];
      }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < sc.n) issue(it + 1, (it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) db_expand(wD[rg], s, rg);
    if (sc.block(it) * 32 + 32 > kl) mask_padded_keys(s, sc.block(it) * 32, half, kl);      // (wave-uniform: no padding, no cost)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
    // element r <-> key j0 + (r&3) + 8*(r>>2) + 4*half ; query = lane.  Masked / padded entries are -inf in the bias.
    float m4[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
#pragma unroll
    for (int e = 0; e < 16; ++e) m4[e & 3] = fmaxf(m4[e & 3], s[e]);
    float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    {
      // the row's other 16 keys sit in lane ^ 32: v_permlane32_swap hands both halves their partner's value in one VALU
      // instruction (__shfl_xor is a ds_bpermute: an LDS round trip + s_waitcnt lgkmcnt(0) in the block's dependent chain)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * LOG2E;
    }
    if (__builtin_amdgcn_ballot_w64(mx > m_run + LAZY_MAX_SLACK)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == NEG_INF) ? 0.f : m_new));
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
    }
    const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
    float ps4[4] = {0.f, 0.f, 0.f, 0.f};
    U128 pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float p0 = __builtin_amdgcn_exp2f(fmaf(s[s2 * 8 + e], LOG2E, -m_use));
        float p1 = __builtin_amdgcn_exp2f(fmaf(s[s2 * 8 + e + 1], LOG2E, -m_use));
        ps4[(e >> 1) & 3] += p0 + p1;                     // the denominator is the undropped row sum
        if (DROP) {
          const int r = s2 * 8 + e, j = sc.block(it) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          p0 *= attn_keep(dr, rk, j); p1 *= attn_keep(dr, rk, j + 1);
        }
        pf[s2].w[e >> 1] = pack2bf(p0, p1);
      }
    l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
    // O^T += V^T P^T ; slot (kh, e) <-> key 16*s2 + 4*kh + (e&3) + 8*(e>>2)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[s2][db].b, pf[s2].b, oacc[db], 0, 0, 0);
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * (a.gain ? a.gain[h] : 1.f);
  if (bact) {
    bf16_t* op = a.out + (long long)b * a.o_bs + (long long)qrow * a.ldo + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) store_tile_bf16(op + db * 32, oacc[db], inv, half, qvalid);
    if (qvalid && half == 0) a.lse_out[((long long)b * a.H + h) * a.T + qi] = m_run + __log2f(l_tot);
  }
}
__global__ __launch_bounds__(512, 2) void attn_bi_fwd_kernel(BiArgs a) { attn_bi_fwd_body<false>(a); }
__global__ __launch_bounds__(512, 2) void attn_bi_fwd_drop_kernel(BiArgs a) { attn_bi_fwd_body<true>(a); }

// ---------------------------------------------------------------------------------------------- dQ (+ sum_b dS)
template <bool DROP>
__device__ __forceinline__ void attn_bi_dq_body(const BiArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* slots = smem + 2 * STG_DQ;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wv >> 2, bl = wv & 3;
  const int nqt = (a.T + 63) >> 6, nbg = (a.B + 3) >> 2;
  // batch group fastest: the workgroups that read the same bias rows run side by side on one XCD (second fetch = L2 hit)
  const int bid = xcd_remap(blockIdx.x, nqt * a.H * nbg);
  const int bg = bid % nbg, h = (bid / nbg) / nqt;
  int qt = (bid / nbg) % nqt;
  if (a.causal) qt = nqt - 1 - qt;                       // later query tiles see more keys: long workgroups first
  const int q0 = qt * 64;
  const int b = bg * 4 + bl;
  const bool bact = b < a.B;
  const int bc = bact ? b : a.B - 1;
  const int qi = q0 + qb * 32 + (lane & 31);
  const bool qvalid = qi < a.T;
  const int qrow = qvalid ? qi : a.T - 1;
  const float gain = a.gain ? a.gain[h] : 1.f;
  const int kl = a.kv_len ? __builtin_amdgcn_readfirstlane(a.kv_len[bc]) : 0x7fffffff;
  AttnDrop dr{};
  unsigned rk = 0;
  if (DROP) { dr = attn_drop_setup(a.drop_p, a.drop_seed, a.drop_seed_add); rk = attn_row_key(dr, ((unsigned)bc * a.H + h) * a.T + qi); }

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = a.q + (long long)bc * a.q_bs + (long long)qrow * a.ldq + h * 64 + half * 8;
    const bf16_t* op = a.dO + (long long)bc * a.do_bs + (long long)qrow * a.lddo + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      U128 u; u.v = *reinterpret_cast<const uint4*>(qp + ks * 16); qf[ks] = u.b;
      U128 w; w.v = *reinterpret_cast<const uint4*>(op + ks * 16); dof[ks] = w.b;
    }
  }
  // an inactive wave (batch element past B) or an invalid query row has lse = +inf: P = 0, dS = 0 -- it still feeds a
  // (zero) tile to the batch sum and helps staging
  const float nlse = (qvalid && bact) ? -a.lse[((long long)bc * a.H + h) * a.T + qi] : NEG_INF;      // log2 units
  const float del = (qvalid && bact) ? a.delta[((long long)bc * a.H + h) * a.T + qi] : 0.f;

  Sched sc;
  {
    const int nkb = a.Sp >> 5;
    if (a.causal) {
      const int pb = a.P >> 5;
      sc.g_begin = 0;
      sc.g_end = q0 < a.P ? min(pb, (min(q0 + 63, a.P - 1) >> 5) + 1) : 0;
      sc.t_begin = pb;
      sc.n = sc.g_end + (nkb - pb);
    } else {
      sc.g_begin = 0; sc.g_end = nkb; sc.t_begin = nkb; sc.n = nkb;
    }
  }

  const bf16_t* kb_ = a.k + (long long)bc * a.k_bs + h * 64;
  const bf16_t* vb_ = a.v + (long long)bc * a.v_bs + h * 64;
  const bf16_t* db_ = a.D + (long long)h * a.Tp * a.Sp;
  const unsigned lds0 = lds_addr(smem);
  // staging of block `it` into stage st: the four 1-KiB pieces of ONE operand tile of this wave's batch element (group 0:
  // V, group 1: K) and a quarter of this group's bias tile.  Lane l of a piece = row 8 p + (l >> 3), 16-byte position
  // l & 7, which holds source chunk (l & 7) ^ swizzle(row).
  const int ldkv = qb ? a.ldk : a.ldv;
  const bi_v4i32 rsKV = bi_rsrc(qb ? kb_ : vb_, (unsigned)(((long long)(a.S - 1) * ldkv + 64) * 2));
  const bi_v4i32 rsD = bi_rsrc(db_, (unsigned)((long long)a.Tp * a.Sp * 2));
  // (three loop-invariant per-lane offsets: this kernel has the registers -- 176 of 256 at two waves per SIMD)
  unsigned ve, vo;
  tile_voff(lane, ldkv, ve, vo);
  const unsigned vD = db_voff(bl & 1, q0 + qb * 32, a.T - 1, a.Sp, 0, lane);
  auto issue = [&](int it, int st) {
    const int j0 = sc.block(it) * 32;
    const unsigned base = lds0 + st * STG_DQ;
    const unsigned dst = base + (qb ? ST_A : ST_B) + bl * 4096;
    const unsigned s0 = (unsigned)(j0 * ldkv) * 2u, sp = (unsigned)ldkv * 16u;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) lds_dma16_bs(rsKV, dst + piece * 1024, (piece & 1) ? vo : ve, s0 + piece * sp);
    if (bl < 2) lds_dma16_bs(rsD, base + ST_D + qb * 2048 + bl * 1024, vD, (unsigned)j0 * 2u);
  };

  f32x16 dq[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq[0][e] = 0.f; dq[1][e] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { consume_frag(qf[ks]); consume_frag(dof[ks]); }

  // per-lane LDS offsets (loop invariant): row reads at chunk half + 2 ks, transposed reads of the K tile
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  int oR[4], oT[2][2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) oR[ks] = vx_off(lane & 31, half * 16) ^ (ks << 5);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int colb = (db * 32 + g16 * 16 + (i16 & 3) * 4) * 2, r0 = 16 * s2 + 4 * half + (i16 >> 2);
      oT[s2][db][0] = vx_off(r0, colb); oT[s2][db][1] = vx_off(r0 + 8, colb);
    }
  const int oD = db_off(lane & 31, half * 8);            // bias seeds of register group rg: ^ (rg << 4)
  const int oS = (lane & 31) * 128 + ((half ^ (lane & 7)) << 4);       // this wave's dS tile: row = query, chunk (half + 2 rg) ^ (row & 7)

  // quarter bl of query block qb's tile: row 8 bl + (lane >> 3), columns 4 (lane & 7) .. + 3
  const int rrow = bl * 8 + (lane >> 3);
  const int rq = q0 + qb * 32 + rrow;
  bf16_t* dbp = a.dbias + (long long)bg * a.dbias_gs + ((long long)h * a.T + (rq < a.T ? rq : a.T - 1)) * a.Sp + 4 * (lane & 7);
  const int oQ = qb * 4 * SLOT + rrow * 128 + ((((lane & 7) ^ (rrow & 7))) << 4);
  auto reduce = [&](int itp) {
    const unsigned char* sl = slots + (itp & 1) * 8 * SLOT + oQ;
    const float4 x0 = *reinterpret_cast<const float4*>(sl), x1 = *reinterpret_cast<const float4*>(sl + SLOT);
    const float4 x2 = *reinterpret_cast<const float4*>(sl + 2 * SLOT), x3 = *reinterpret_cast<const float4*>(sl + 3 * SLOT);
    const float s0 = (x0.x + x1.x) + (x2.x + x3.x), s1 = (x0.y + x1.y) + (x2.y + x3.y);
    const float s2 = (x0.z + x1.z) + (x2.z + x3.z), s3 = (x0.w + x1.w) + (x2.w + x3.w);
    if (rq < a.T) *reinterpret_cast<uint2*>(dbp + sc.block(itp) * 32) = make_uint2(pack2bf(s0, s1), pack2bf(s2, s3));
  };

  float pdp = 0.f;
  // One 32-key block: every LDS read of the block is requested up front (bias seeds, K rows, V rows, the transposed K
  // fragments of the dQ MFMAs: 20 requests), then S^T = bias + K Q^T and dP^T = V dO^T, the exp / dS segment, this wave's dS
  // tile into its slot of the batch sum, dQ^T += K^T dS^T.
  auto block = [&](int it) {
    const unsigned char* stg = smem + (it & 1) * STG_DQ;
    const unsigned char* sK = stg + ST_A + bl * 4096;
    const unsigned char* sV = stg + ST_B + bl * 4096;
    const unsigned char* sD = stg + ST_D + qb * 2048;
    f32x16 s, dp;
    bf16x8 kf[4], vf[4];
    U128 f[2][2];
    uint2 wD[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) wD[rg] = *reinterpret_cast<const uint2*>(sD + (oD ^ (rg << 4)));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = lds_read_b128(sK + oR[ks]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = lds_read_b128(sV + oR[ks]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        U64 x, y;
        x.s = lds_read_tr(sK + oT[s2][db][0]);
        y.s = lds_read_tr(sK + oT[s2][db][1]);
        f[s2][db].w[0] = x.w[0]; f[s2][db].w[1] = x.w[1]; f[s2][db].w[2] = y.w[0]; f[s2][db].w[3] = y.w[1];
      }
    __builtin_amdgcn_sched_barrier(0);
    // the next block's staging is issued HERE: the ~5 LDS-DMA pieces of a wave cost several hundred issue cycles, which now
    // pass while this block's LDS reads return (the other stage is free since the barrier)
    if (it + 1 < sc.n) issue(it + 1, (it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) db_expand(wD[rg], s, rg);
    if (sc.block(it) * 32 + 32 > kl) mask_padded_keys(s, sc.block(it) * 32, half, kl);      // (wave-uniform: no padding, no cost)
#pragma unroll
    for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], dof[ks], dp, 0, 0, 0);
    }
    // element r <-> key j0 + (r&3) + 8*(r>>2) + 4*half ; query = lane
    float ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[e], LOG2E, nlse));
      float dpe = dp[e];
      if (DROP) dpe *= attn_keep(dr, rk, sc.block(it) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half);
      ds[e] = p * fmaf(gain, dpe, -del);
      pdp = fmaf(p, dpe, pdp);            // sum_j P_ij dP_ij = dO_i . (P V)_i : the c_attn gradient without dividing by c_attn
    }
    unsigned char* sl = slots + ((it & 1) * 8 + wv) * SLOT;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      *reinterpret_cast<float4*>(sl + (oS ^ (rg << 5))) = make_float4(ds[rg * 4], ds[rg * 4 + 1], ds[rg * 4 + 2], ds[rg * 4 + 3]);
    U128 ud[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; e += 2) ud[s2].w[e >> 1] = pack2bf(ds[s2 * 8 + e], ds[s2 * 8 + e + 1]);
    // dQ^T += K^T dS^T ; slot (kh, e) <-> key 16*s2 + 4*kh + (e&3) + 8*(e>>2)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s2][db].b, ud[s2].b, dq[db], 0, 0, 0);
  };

  if (sc.n > 0) issue(0, 0);
  for (int it = 0; it < sc.n; ++it) {
    lds_dma_wait();
    __syncthreads();                 // stage `it` has landed; every dS tile of block it-1 is in its slot
    if (it > 0) reduce(it - 1);
    block(it);
  }
  if (sc.n > 0) {
    __syncthreads();
    reduce(sc.n - 1);
  }
  if (a.dgain_rows) {
    // d c_attn[h] = sum_{b,t} dO[b,t,h,:] . O_pre[b,t,h,:], O_pre = P V (before the gain): the row sums leave here, exact for
    // every value of c_attn (delta / c_attn is 0 / 0 at c_attn = 0)
    const float tot = pdp + __shfl_xor(pdp, 32);
    if (bact && qvalid && half == 0) a.dgain_rows[((long long)b * a.H + h) * a.T + qi] = tot;
  }
  if (bact) {
    bf16_t* dqp = a.dq + (long long)b * a.dq_bs + (long long)qrow * a.lddq + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) store_tile_bf16(dqp + db * 32, dq[db], a.dq_scale, half, qvalid);
  }
}
__global__ __launch_bounds__(512, 2) void attn_bi_dq_kernel(BiArgs a) { attn_bi_dq_body<false>(a); }
__global__ __launch_bounds__(512, 2) void attn_bi_dq_drop_kernel(BiArgs a) { attn_bi_dq_body<true>(a); }

// ---------------------------------------------------------------------------------------------- dK / dV
template <bool DROP>
__device__ __forceinline__ void attn_bi_dkv_body(const BiArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kbw = wv >> 2, bl = wv & 3;
  const int nkt = (a.S + 63) >> 6, nbg = (a.B + 3) >> 2;
  const int bid = xcd_remap(blockIdx.x, nkt * a.H * nbg);
  const int bg = bid % nbg, kt = (bid / nbg) % nkt, h = (bid / nbg) / nkt;
  const int k0 = kt * 64;
  const int b = bg * 4 + bl;
  const bool bact = b < a.B;
  const int bc = bact ? b : a.B - 1;
  const int kj = k0 + kbw * 32 + (lane & 31);
  const bool kvalid = kj < a.S;
  const int krow = kvalid ? kj : a.S - 1;
  const float gain = a.gain ? a.gain[h] : 1.f;
  // key padding: this lane's key is beyond the batch element's valid count (all of its 16 scores are masked); `anypad`: the
  // wave's key block touches the padding at all (wave-uniform)
  const int kl = a.kv_len ? __builtin_amdgcn_readfirstlane(a.kv_len[bc]) : 0x7fffffff;
  const bool kpad = kj >= kl, anypad = k0 + kbw * 32 + 32 > kl;
  AttnDrop dr{};
  if (DROP) dr = attn_drop_setup(a.drop_p, a.drop_seed, a.drop_seed_add);
  const unsigned rowbase = ((unsigned)bc * a.H + h) * a.T;

  bf16x8 kf[4], vfn[4];
  {
    const bf16_t* kp = a.k + (long long)bc * a.k_bs + (long long)krow * a.ldk + h * 64 + half * 8;
    const bf16_t* vp = a.v + (long long)bc * a.v_bs + (long long)krow * a.ldv + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      U128 u; u.v = *reinterpret_cast<const uint4*>(kp + ks * 16); kf[ks] = u.b;
      // -gain * V: with the dP accumulator seeded with delta the MFMAs leave delta - gain * dP = -(dS / P)
      U128 w; w.v = scale_bf16x8(*reinterpret_cast<const uint4*>(vp + ks * 16), -gain); vfn[ks] = w.b;
    }
  }
  Sched sc;
  {
    const int nqb = a.Tp >> 5;
    if (a.causal) {
      const int pb = a.P >> 5;
      if (k0 < a.P) {     // grid keys: visible to the grid rows at or after them, to no tail row
        sc.g_begin = k0 >> 5; sc.g_end = pb; sc.t_begin = nqb; sc.n = sc.g_end - sc.g_begin;
      } else {            // tail keys: every grid row, and the tail rows (masked element-wise inside the bias)
        sc.g_begin = 0; sc.g_end = pb; sc.t_begin = pb; sc.n = nqb;
      }
    } else {
      sc.g_begin = 0; sc.g_end = nqb; sc.t_begin = nqb; sc.n = nqb;
    }
  }
  const bf16_t* qb_ = a.q + (long long)bc * a.q_bs + h * 64;
  const bf16_t* ob_ = a.dO + (long long)bc * a.do_bs + h * 64;
  const float* lb_ = a.lse + ((long long)bc * a.H + h) * a.T;
  const float* eb_ = a.delta + ((long long)bc * a.H + h) * a.T;
  const bf16_t* db_ = a.D + (long long)h * a.Tp * a.Sp;
  const unsigned lds0 = lds_addr(smem);
  // group 0: the Q tile, group 1: the dO tile of this wave's batch element (rows 0 .. T-1, 64 columns of head h)
  const int ldqo = kbw ? a.lddo : a.ldq;
  const bi_v4i32 rsQO = bi_rsrc(kbw ? ob_ : qb_, (unsigned)(((long long)(a.T - 1) * ldqo + 64) * 2));
  const bi_v4i32 rsD = bi_rsrc(db_, (unsigned)((long long)a.Tp * a.Sp * 2));
  unsigned ve, vo;                                            // (loop-invariant per-lane offsets, see tile_voff / db_voff)
  tile_voff(lane, ldqo, ve, vo);
  const int jc = min(k0 + kbw * 32, a.Sp - 32);              // (a key block entirely in the padding: any valid address)
  const unsigned vD = db_voff(bl & 1, 0, 0x7fffffff, a.Sp, jc, lane);
  auto issue = [&](int it, int st) {
    const int i0 = sc.block(it) * 32;
    const unsigned base = lds0 + st * STG_DKV;
    const unsigned dst = base + (kbw ? ST_B : ST_A) + bl * 4096;
    const unsigned s0 = (unsigned)(i0 * ldqo) * 2u, sp = (unsigned)ldqo * 16u;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) lds_dma16_bs(rsQO, dst + piece * 1024, (piece & 1) ? vo : ve, s0 + piece * sp);
    if (bl < 2) {
      // bias tile [32 queries][32 keys of key block kbw].  The accumulators want four CONSECUTIVE QUERIES of one key per lane --
      // a column of this tile: read with the hardware transpose (ds_read_b64_tr_b16), four 8-byte reads per block.  (The
      // fp32 tile of round 4 served them with sixteen 4-byte reads; before that a second, transposed copy of the bias in HBM.)
      lds_dma16_bs(rsD, base + ST_D + kbw * 2048 + bl * 1024, vD, (unsigned)(i0 * a.Sp) * 2u);
    }
    if (kbw == 0) {
      // lanes 0..31: lse, lanes 32..63: delta of this wave's batch element (LDS-DMA places lane i at base + 4 i)
      const int ir = min(i0 + (lane & 31), a.T - 1);
      if (lane < 32) lds_dma4_gs(lb_, ir * 4, base + ST_L + bl * 256);
      else lds_dma4_gs(eb_, ir * 4, base + ST_L + bl * 256);
    }
  };

  f32x16 dv[2], dk[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { dv[0][e] = 0.f; dv[1][e] = 0.f; dk[0][e] = 0.f; dk[1][e] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { consume_frag(kf[ks]); consume_frag(vfn[ks]); }

  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  int oR[4], oT[2][2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) oR[ks] = vx_off(lane & 31, half * 16) ^ (ks << 5);
  // bias seeds: register group rg <-> queries 8 rg + 4 half .. + 3 of key (lane & 31): lane i of a 16-lane group passes the
  // address of the 8-byte piece (row i >> 2, keys 4 (i & 3) ..) of the [4 queries][16 keys] block and receives its column i
  int oDt[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) oDt[rg] = db_off(8 * rg + 4 * half + (i16 >> 2), (g16 * 16 + (i16 & 3) * 4) * 2);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int colb = (db * 32 + g16 * 16 + (i16 & 3) * 4) * 2, r0 = 16 * s2 + 4 * half + (i16 >> 2);
      oT[s2][db][0] = vx_off(r0, colb); oT[s2][db][1] = vx_off(r0 + 8, colb);
    }

  // One 32-row block of the streamed side.  EVERY LDS read of the block is requested up front -- bias seeds, delta seeds,
  // Q / dO rows, row statistics and the transposed Q / dO fragments of the gradient MFMAs, 40 requests -- so a wave sleeps on
  // LDS latency once per block instead of at five dependent points (ablation, profiles/round4_attn_bi_ablation.txt: with every MFMA and the
  // staging removed the v2 loop still took 68 of its 121 us: two waves per SIMD do not hide ~5 x 150 cycles of dependent LDS
  // round trips per block).  (Measured and dropped: the gradient MFMAs of block n-1 under the row reads of block n, 110.7 vs
  // 106.2 us.)
  auto block = [&](int it) {
    const int st = it & 1;
    const unsigned char* stg = smem + st * STG_DKV;
    const unsigned char* sQ = stg + ST_A + bl * 4096;
    const unsigned char* sO = stg + ST_B + bl * 4096;
    const unsigned char* sD = stg + ST_D + kbw * 2048;
    const float* sL = reinterpret_cast<const float*>(stg + ST_L + bl * 256);
    f32x16 s, dp;
    uint2 wD[4];
    float ls[16], de[DROP ? 16 : 1];
    bf16x8 qf[4], of[4];
    U128 fo[2][2], fq[2][2];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      {
        int od = oDt[rg];
        if (DROP) {       // (the dropping instantiation has no register left for four hoisted offsets: re-derived per block)
          int ll = lane;
          asm volatile("" : "+v"(ll));
          od = db_off(8 * rg + 4 * (ll >> 5) + ((ll & 15) >> 2), (((ll >> 4) & 1) * 16 + (ll & 3) * 4) * 2);
        }
        U64 w; w.s = lds_read_tr(sD + od); wD[rg] = w.v;
      }
      const float4 e4 = *reinterpret_cast<const float4*>(sL + 32 + 8 * rg + 4 * half);
      dp[rg * 4] = e4.x; dp[rg * 4 + 1] = e4.y; dp[rg * 4 + 2] = e4.z; dp[rg * 4 + 3] = e4.w;
      if (DROP) { de[rg * 4] = e4.x; de[rg * 4 + 1] = e4.y; de[rg * 4 + 2] = e4.z; de[rg * 4 + 3] = e4.w; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = lds_read_b128(sQ + oR[ks]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) of[ks] = lds_read_b128(sO + oR[ks]);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 l4 = *reinterpret_cast<const float4*>(sL + 8 * rg + 4 * half);
      ls[rg * 4] = l4.x; ls[rg * 4 + 1] = l4.y; ls[rg * 4 + 2] = l4.z; ls[rg * 4 + 3] = l4.w;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        U64 x, y;
        x.s = lds_read_tr(sO + oT[s2][db][0]);
        y.s = lds_read_tr(sO + oT[s2][db][1]);
        fo[s2][db].w[0] = x.w[0]; fo[s2][db].w[1] = x.w[1]; fo[s2][db].w[2] = y.w[0]; fo[s2][db].w[3] = y.w[1];
        x.s = lds_read_tr(sQ + oT[s2][db][0]);
        y.s = lds_read_tr(sQ + oT[s2][db][1]);
        fq[s2][db].w[0] = x.w[0]; fq[s2][db].w[1] = x.w[1]; fq[s2][db].w[2] = y.w[0]; fq[s2][db].w[3] = y.w[1];
      }
    __builtin_amdgcn_sched_barrier(0);
    // (the next block's staging is issued while this block's LDS reads return: see the dQ kernel)
    if (it + 1 < sc.n) issue(it + 1, (it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) db_expand(wD[rg], s, rg);
    if (anypad) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = kpad ? NEG_INF : s[e];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[ks], vfn[ks], dp, 0, 0, 0);
    }
    // element r <-> query i0 + (r&3) + 8*(r>>2) + 4*half ; key = lane
    U128 up[2], ud[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int r = s2 * 8 + e;
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -ls[r]));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], LOG2E, -ls[r + 1]));
        if (DROP) {
          // the accumulator holds delta - gain dP: with the mask, -(dS / P) = delta - keep / (1 - p) * gain dP
          const int i = sc.block(it) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float k0_ = attn_keep(dr, attn_row_key(dr, rowbase + i), kj), k1_ = attn_keep(dr, attn_row_key(dr, rowbase + i + 1), kj);
          up[s2].w[e >> 1] = pack2bf(p0 * k0_, p1 * k1_);
          ud[s2].w[e >> 1] = pack2bf(-p0 * (de[r] - k0_ * (de[r] - dp[r])), -p1 * (de[r + 1] - k1_ * (de[r + 1] - dp[r + 1])));
        } else {
          up[s2].w[e >> 1] = pack2bf(p0, p1);
          ud[s2].w[e >> 1] = pack2bf(-p0 * dp[r], -p1 * dp[r + 1]);
        }
      }
    // dV^T += dO^T P ; dK^T += Q^T dS ; slot (kh, e) <-> query 16*s2 + 4*kh + (e&3) + 8*(e>>2)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int db = 0; db < 2; ++db) dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[s2][db].b, up[s2].b, dv[db], 0, 0, 0);
#pragma unroll
      for (int db = 0; db < 2; ++db) dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s2][db].b, ud[s2].b, dk[db], 0, 0, 0);
    }
  };

  // (Measured and dropped: staging two blocks ahead through a ring of three stages -- 112.2 vs 106.2 us.)
  if (sc.n > 0) issue(0, 0);
  for (int it = 0; it < sc.n; ++it) {
    lds_dma_wait();
    __syncthreads();                   // block `it` has landed; everyone is done with block it-1's tiles
    block(it);
  }
  if (bact && kvalid) {
    bf16_t* dvp = a.dv + (long long)b * a.dv_bs + (long long)kj * a.lddv + h * 64;
    bf16_t* dkp = a.dk + (long long)b * a.dk_bs + (long long)kj * a.lddk + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      store_tile_bf16(dvp + db * 32, dv[db], gain, half, true);
      store_tile_bf16(dkp + db * 32, dk[db], 1.f, half, true);
    }
  }
}
__global__ __launch_bounds__(512, 2) void attn_bi_dkv_kernel(BiArgs a) { attn_bi_dkv_body<false>(a); }
__global__ __launch_bounds__(512, 2) void attn_bi_dkv_drop_kernel(BiArgs a) { attn_bi_dkv_body<true>(a); }

// the mask the three kernels apply, written out (tests; a debugging aid): keep[b, h, i, j] in {0, 1}
__global__ void attn_drop_mask_kernel(unsigned char* out, int B, int H, int T, int S, float p, unsigned long long seed,
                                      const unsigned long long* seed_add) {
  const long long n = (long long)B * H * T * S, id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  const AttnDrop dr = attn_drop_setup(p, seed, seed_add);
  const int j = (int)(id % S);
  const unsigned row = (unsigned)(id / S);            // (b * H + h) * T + i
  out[id] = attn_keep(dr, attn_row_key(dr, row), j) != 0.f;
}

// ---------------------------------------------------------------------------------------------- dense bias
struct DenseArgs {
  const bf16_t *pq, *pk;
  int ldpq, ldpk, H, T, S, Sp, Tp;
  int rel_mode, P, code_bias, n2d, Lt, causal;
  const int* gcode;
  const float *rel2d, *rel1d, *relx;
  bf16_t* D;
};

// D[h][i][j] = pos_q[i] . pos_k[j] + rel(i, j) as [H][Tp][Sp], -inf where (i, j) is masked (causal, "tail-first" order),
// j >= S or i >= T.  One workgroup per (head, 32-row strip): the head's delta table and the grid codes sit in LDS, each
// wave walks 32 x 32 tiles of the strip (accumulator row = the lane: a tile leaves as 16-byte row segments).
__global__ __launch_bounds__(512) void attn_dense_bias_kernel(DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sT = reinterpret_cast<float*>(smem);                       // rel2d[h]
  float* s1 = sT + ((a.n2d + 3) & ~3);                              // rel1d[h]
  int* sG = reinterpret_cast<int*>(s1 + ((2 * a.Lt + 2) & ~3));     // gcode
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, x = lane & 31;
  const int ntq = a.Tp >> 5, ntk = a.Sp >> 5;
  const int h = blockIdx.x / ntq, ti = blockIdx.x % ntq, i0 = ti * 32;
  float rx0 = 0.f, rx1 = 0.f;
  if (a.rel_mode) {
    for (int i = tid; i < a.n2d; i += 512) sT[i] = a.rel2d[(long long)h * a.n2d + i];
    for (int i = tid; i < 2 * a.Lt - 1; i += 512) s1[i] = a.rel1d[(long long)h * (2 * a.Lt - 1) + i];
    for (int i = tid; i < a.P; i += 512) sG[i] = a.gcode[i];
    rx0 = a.relx[h * 2]; rx1 = a.relx[h * 2 + 1];
  }
  __syncthreads();
  auto entry = [&](int i, int j, float abs_ij) -> float {
    if (i >= a.T || j >= a.S) return NEG_INF;
    if (a.causal) {
      const bool masked = (j < a.P) ? ((i >= a.P) || (j > i)) : ((i >= a.P) && (j > i));
      if (masked) return NEG_INF;
    }
    float r = 0.f;
    if (a.rel_mode) {
      if (i < a.P) r = (j < a.P) ? sT[sG[i] - sG[j] + a.code_bias] : rx0;
      else r = (j < a.P) ? rx1 : s1[(i - j) + a.Lt - 1];
    }
    return abs_ij + r;
  };
  bf16x8 fq[4];
  if (a.pq) {
    const bf16_t* qp = a.pq + (long long)min(i0 + x, a.T - 1) * a.ldpq + h * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { U128 u; u.v = *reinterpret_cast<const uint4*>(qp + ks * 16); fq[ks] = u.b; }
  }
  for (int tj = wave; tj < ntk; tj += 8) {
    const int j0 = tj * 32;
    // (causal) a tile entirely above the diagonal: -inf without any look-up; one that neither backward kernel's block schedule
    // ever reaches (they walk 64-row tiles: a 32-row block reads at most one 32-column block beyond its own diagonal block)
    // is not written at all
    const bool dead = a.causal && j0 < a.P && (i0 >= a.P || j0 > i0 + 31);
    if (dead && (i0 >= a.P || j0 > i0 + 63)) continue;
    bf16x8 fk[4];
    if (a.pq && !dead) {
      const bf16_t* kp = a.pk + (long long)min(j0 + x, a.S - 1) * a.ldpk + h * 64 + half * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { U128 w; w.v = *reinterpret_cast<const uint4*>(kp + ks * 16); fk[ks] = w.b; }
    }
    f32x16 acc;
    // lane = key j0 + x, element r <-> query i0 + (r&3) + 8*(r>>2) + 4*half (the other orientation -- a lane owning a query
    // row, 16-byte stores -- put 32-byte pieces of 32 rows into every store and measured 1.3 x the algorithmic HBM write
    // bytes, profiles/round4_hbm_traffic.json)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (a.pq && !dead) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks], fk[ks], acc, 0, 0, 0);
    }
    {
      // bf16 pairs of neighbouring keys: lanes x and x ^ 1 exchange two of their four rows per register group (DPP quad_perm
      // [1,0,3,2]) -- the even lane stores rows e = 0, 1 of the key pair, the odd lane rows e = 2, 3 -- so a store instruction
      // writes 64-byte row pieces of four rows (dwords), not 2-byte elements
      unsigned* dp = reinterpret_cast<unsigned*>(a.D + ((long long)h * a.Tp + i0 + 4 * half) * a.Sp + j0 + (x & ~1));
      const bool odd = x & 1;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dead ? NEG_INF : entry(i0 + 8 * rg + 4 * half + e, j0 + x, acc[rg * 4 + e]);
        const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];
        const float r0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xf, 0xf, true));
        const float r1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xf, 0xf, true));
        const unsigned w0 = odd ? pack2bf(r0, v[2]) : pack2bf(v[0], r0), w1 = odd ? pack2bf(r1, v[3]) : pack2bf(v[1], r1);
        const int e0 = odd ? 2 : 0;
        dp[((long long)(8 * rg + e0) * a.Sp) >> 1] = w0;
        dp[((long long)(8 * rg + e0 + 1) * a.Sp) >> 1] = w1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- gradients of the bias
constexpr int DB_NSUB = 8;            // Toeplitz-table blocks per (head, part)
constexpr int DB_NPARTS = 4;          // partial delta tables per head (summed by ifseg_attn_bwd_reduce in a fixed order)
struct DbArgs {
  const bf16_t* dbias;        // [ng][H][T][Sp]
  long long gs;               // elements per group slab
  int ng, H, T, S, Sp, C;
  const bf16_t *pq, *pk;      // [T, ldpq], [S, ldpk]
  int ldpq, ldpk;
  float *dpq, *dpk;           // fp32 [T, C], [S, C]
  int accumulate;             // operand gradients: add to what dpq / dpk hold
  int tab_accumulate;         // partial delta tables: add to what they hold (slab pairs after the first)
  float dpq_scale;
  int P, gh, gw, Lt, causal;
  float *drel2d, *drel1d, *drelx;     // [H][DB_NPARTS][(2gh-1)(2gw-1)], [H][DB_NPARTS][2Lt-1], [H][DB_NPARTS][2]
  int nb_q, nb_k, nb_2d;
};

// A / B fragment of a transposed read of a [rows = contraction index][128-byte row] tile (vx layout): 16 rows from
// `rbase`, columns cb*32 .. +31 = the fragment's lane index; slot (kh, e) <-> row rbase + 4*kh + (e&3) + 8*(e>>2)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int rbase, int cb, int lane) {
  const int half = lane >> 5, i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int col = cb * 32 + g16 * 16 + (i16 & 3) * 4;
  const int r0 = rbase + 4 * half + (i16 >> 2);
  U64 x, y;
  x.s = lds_read_tr(tile + vx_off(r0, col * 2));
  y.s = lds_read_tr(tile + vx_off(r0 + 8, col * 2));
  U128 f; f.w[0] = x.w[0]; f.w[1] = x.w[1]; f.w[2] = y.w[0]; f.w[3] = y.w[1];
  return f.b;
}

// Two block ranges (d pos_q | d pos_k): small GEMMs with the contraction split over the four waves of a workgroup
// (each wave stages its own tiles in its own LDS region: no block barrier inside the loop; the next step's global loads
// are in flight under the current step's MFMAs) and a fixed-order sum of the four accumulators at the end.
__global__ __launch_bounds__(256) void attn_dbias_grads_kernel(DbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // consecutive tiles of a head on ONE XCD, next to each other in time: a d pos_k block reads 64 bytes of every row and its
  // neighbour the other half of the same 128-byte lines (round-robin placement measured 1.75 x the algorithmic read bytes)
  int blk = blockIdx.x < a.nb_q ? xcd_remap(blockIdx.x, a.nb_q) : a.nb_q + xcd_remap(blockIdx.x - a.nb_q, a.nb_k);
  const int lr = lane >> 3, lc = lane & 7;        // tile staging: rows lr + 8 u, 16-byte chunk lc
  if (blk < a.nb_q) {
    // ---- d pos_q[i][h*64 + c] (+)= scale * sum_j dB[h][i][j] pos_k[j][h*64 + c]:  out^T[c][i], A = pos_k^T (tr), B = dB rows
    const int nit = (a.T + 31) >> 5;
    const int h = blk / nit, i0 = (blk % nit) * 32;
    const int i = i0 + (lane & 31);
    const int ir = i < a.T ? i : a.T - 1;
    unsigned char* tile = sm + wave * 8192;       // pos_k tile [32 j][64 c]; behind it the dB tile [32 i][ng x 32 j] (64-byte halves)
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    uint4 t4[4], d4[4];
    // both tiles arrive as whole 64-byte row pieces (16 bytes per lane) and go through LDS; the B fragments -- a lane's
    // eight dB values of ITS row in the k-slot order of the transposed A read -- are two 8-byte LDS reads each.  (Fetched
    // straight into fragment shape, a wave instruction took 16 bytes out of each of 32 rows: a quarter of every 64-byte
    // segment it touched, four times over.)
    auto fetch = [&](int j0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {          // pos_k tile [32 j][64 c], rows past S are zero
        t4[u] = make_uint4(0, 0, 0, 0); d4[u] = make_uint4(0, 0, 0, 0);
        if (j0 + lr + 8 * u < a.S) t4[u] = *reinterpret_cast<const uint4*>(a.pk + (long long)(j0 + lr + 8 * u) * a.ldpk + h * 64 + lc * 8);
        const int g = lc >> 2, irow = min(i0 + lr + 8 * u, a.T - 1);
        if (g < a.ng) d4[u] = *reinterpret_cast<const uint4*>(a.dbias + g * a.gs + ((long long)h * a.T + irow) * a.Sp + j0 + (lc & 3) * 8);
      }
    };
    // (causal: sum_b dS is zero -- and was never written -- for the grid columns beyond the block's last row, and for every
    // grid column when the rows are tail rows; the schedule of the dQ kernel, step by step)
    auto live = [&](int j) { return !(a.causal && j < a.P && (i0 >= a.P || j > i0 + 31)); };
    int j0 = wave * 32;
    if (j0 < a.Sp) fetch(j0);
    for (; j0 < a.Sp; j0 += 128) {
      if (!live(j0)) { if (j0 + 128 < a.Sp) fetch(j0 + 128); continue; }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = vx_off(lr + 8 * u, lc * 16);
        *reinterpret_cast<uint4*>(tile + o) = t4[u];
        *reinterpret_cast<uint4*>(tile + 4096 + o) = d4[u];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (j0 + 128 < a.Sp) fetch(j0 + 128);
      U128 bfr[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          // row = the lane's query, columns j0 + 16 s2 + 4 half .. + 3 and the same + 8 (group g's 64-byte half of the row)
          const int cb = g * 64 + (16 * s2 + 4 * half) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(tile + 4096 + vx_off(lane & 31, cb));
          const uint2 hi = *reinterpret_cast<const uint2*>(tile + 4096 + vx_off(lane & 31, cb + 16));
          bfr[s2][g].w[0] = lo.x; bfr[s2][g].w[1] = lo.y; bfr[s2][g].w[2] = hi.x; bfr[s2][g].w[3] = hi.y;
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 a0 = tr_frag(tile, 16 * s2, 0, lane), a1 = tr_frag(tile, 16 * s2, 1, lane);
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (g < a.ng) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr[s2][g].b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfr[s2][g].b, acc[1], 0, 0, 0);
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                   // (the tiles are dead: the sum reuses their LDS)
    // fixed-order sum of the four waves' accumulators, one column block at a time: [wave][r][lane]
    float* red = reinterpret_cast<float*>(sm);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      if (cb) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[cb][r];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + 256 * k, r = v >> 6, ln = v & 63;
        const float t = (red[r * 64 + ln] + red[(16 + r) * 64 + ln]) + (red[(32 + r) * 64 + ln] + red[(48 + r) * 64 + ln]);
        const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), ii = i0 + (ln & 31);
        if (ii < a.T) {
          float* p = a.dpq + (long long)ii * a.C + h * 64 + c;
          *p = (a.accumulate ? *p : 0.f) + t * a.dpq_scale;
        }
      }
    }
    return;
  }
  blk -= a.nb_q;
  {
    // ---- d pos_k[j][h*64 + c] (+)= sum_i dB[h][i][j] pos_q[i][h*64 + c]:  out^T[c][j], A = pos_q^T (tr), B = dB^T (tr)
    // 32 columns j per workgroup
    const int njt = a.Sp >> 5;
    const int h = blk / njt, jw0 = (blk % njt) * 32;
    unsigned char* sQ = sm + wave * 8192;       // pos_q tile [32 i][64 c]; behind it the dB tile [32 i][ng x 32 j] (64-byte halves)
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    uint4 t4[4], d4[4];
    auto fetch = [&](int i0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = lr + 8 * u;
        t4[u] = make_uint4(0, 0, 0, 0); d4[u] = make_uint4(0, 0, 0, 0);
        if (i0 + r < a.T) {
          t4[u] = *reinterpret_cast<const uint4*>(a.pq + (long long)(i0 + r) * a.ldpq + h * 64 + lc * 8);
          // chunks 0..3: group 0's 32 columns, chunks 4..7: group 1's
          const int g = lc >> 2;
          if (g < a.ng) d4[u] = *reinterpret_cast<const uint4*>(a.dbias + g * a.gs + ((long long)h * a.T + i0 + r) * a.Sp + jw0 + (lc & 3) * 8);
        }
      }
    };
    auto live = [&](int i) { return !(a.causal && jw0 < a.P && (i >= a.P || jw0 > i + 31)); };
    int i0 = wave * 32;
    if (i0 < a.T) fetch(i0);
    for (; i0 < a.T; i0 += 128) {
      if (!live(i0)) { if (i0 + 128 < a.T) fetch(i0 + 128); continue; }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = vx_off(lr + 8 * u, lc * 16);
        *reinterpret_cast<uint4*>(sQ + o) = t4[u];
        *reinterpret_cast<uint4*>(sQ + 4096 + o) = d4[u];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (i0 + 128 < a.T) fetch(i0 + 128);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 a0 = tr_frag(sQ, 16 * s2, 0, lane), a1 = tr_frag(sQ, 16 * s2, 1, lane);
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (g < a.ng) {
            const bf16x8 bfr = tr_frag(sQ + 4096, 16 * s2, g, lane);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfr, acc[1], 0, 0, 0);
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                   // (the tiles are dead: the sum reuses their LDS)
    float* red = reinterpret_cast<float*>(sm);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      if (cb) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[cb][r];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + 256 * k, r = v >> 6, ln = v & 63;
        const float t = (red[r * 64 + ln] + red[(16 + r) * 64 + ln]) + (red[(32 + r) * 64 + ln] + red[(48 + r) * 64 + ln]);
        const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), j = jw0 + (ln & 31);
        if (j < a.S) {
          float* p = a.dpk + (long long)j * a.C + h * 64 + c;
          *p = (a.accumulate ? *p : 0.f) + t;
        }
      }
    }
  }
}

// delta-table gradients from sum_b dS: block ranges [2-D grid table | scalar entries + Toeplitz table]
__global__ __launch_bounds__(256) void attn_dbias_tables_kernel(DbArgs a) {
  __shared__ __attribute__((aligned(16))) float sf[2 * 4096];
  const int tid = threadIdx.x;
  int blk = blockIdx.x < a.nb_2d ? xcd_remap(blockIdx.x, a.nb_2d) : blockIdx.x;
  auto add8 = [](float* t, const uint4& v) {
    t[0] += bflo(v.x); t[1] += bfhi(v.x); t[2] += bflo(v.y); t[3] += bfhi(v.y);
    t[4] += bflo(v.z); t[5] += bfhi(v.z); t[6] += bflo(v.w); t[7] += bfhi(v.w);
  };
  if (blk < a.nb_2d) {
    // ---- d rel2d[h][part][(dy + gh-1)(2gw-1) + dx + gw-1] = sum over the grid pairs (i, j) with y_i - y_j = dy,
    // x_i - x_j = dx and y_i = part (mod DB_NPARTS).  A thread owns (x_i, 8 consecutive x_j) of every second row: 16-byte loads,
    // all of a thread's loads in flight together.
    const int ndy = 2 * a.gh - 1, w = a.gw, nch = w >> 3, npair = w * nch;
    // (dy fastest, on one XCD: the blocks of dy and dy + 1 read the two halves of the same 128-byte lines on a 32-wide grid)
    const int dy = blk % ndy - (a.gh - 1), part = (blk / ndy) % DB_NPARTS, h = blk / (DB_NPARTS * ndy);
    const int ylo = dy > 0 ? dy : 0, yhi = dy < 0 ? a.gh + dy : a.gh;
    const int y0 = ylo + ((part - ylo) % DB_NPARTS + DB_NPARTS) % DB_NPARTS;
    float* out = a.drel2d + ((long long)h * DB_NPARTS + part) * ndy * (2 * w - 1) + (long long)(dy + a.gh - 1) * (2 * w - 1);
    if (a.causal && dy < 0) {           // pairs with y_j > y_i are masked: nothing was written there
      if (!a.tab_accumulate) for (int d = tid; d < 2 * w - 1; d += 256) out[d] = 0.f;
      return;
    }
    const int nyp = npair <= 128 ? 2 : 1;                       // row phases handled side by side
    const int yp = tid / npair, p = tid - yp * npair;
    for (int p0 = 0; p0 < npair; p0 += 256) {                   // (one pass unless the grid is wider than 40)
      const int pp = p0 + p;
      float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool on = yp < nyp && pp < npair;
      const int xi = on ? pp / nch : 0, xc = on ? pp - xi * nch : 0;
      if (on) {
        const bf16_t* base = a.dbias + ((long long)h * a.T + xi) * a.Sp + xc * 8 - (long long)dy * w;
        const long long ystep = (long long)w * a.Sp + w;        // one grid row down in i and in j
#pragma unroll 4
        for (int yi = y0 + yp * DB_NPARTS; yi < yhi; yi += nyp * DB_NPARTS) {
          add8(t, *reinterpret_cast<const uint4*>(base + yi * ystep));
          if (a.ng > 1) add8(t, *reinterpret_cast<const uint4*>(base + a.gs + yi * ystep));
        }
      }
      if (p0) __syncthreads();
      if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sf[yp * 4096 + xi * w + xc * 8 + e] = t[e];
      }
      __syncthreads();
      // (a grid of more than 256 pairs per phase takes several passes; the diagonal sums below run after the last)
    }
    for (int d = tid; d < 2 * w - 1; d += 256) {
      const int dx = d - (w - 1);
      float t = 0.f;
      for (int xi = (dx > 0 ? dx : 0); xi < (dx < 0 ? w + dx : w); ++xi) {
        t += sf[xi * w + (xi - dx)];
        if (nyp > 1) t += sf[4096 + xi * w + (xi - dx)];
      }
      out[d] = (a.tab_accumulate ? out[d] : 0.f) + t;
    }
    return;
  }
  blk -= a.nb_2d;
  if (blk >= a.H * DB_NPARTS) {
    // ---- (head, part, sub): the tail x tail Toeplitz table, rows i = part (mod DB_NPARTS), diagonals d = sub (mod DB_NSUB);
    // a thread owns a diagonal (i - j = off) and walks its rows in order
    blk -= a.H * DB_NPARTS;
    const int sub = blk % DB_NSUB, part = (blk / DB_NSUB) % DB_NPARTS, h = blk / (DB_NSUB * DB_NPARTS), P = a.P, Lt = a.Lt;
    const bf16_t* hb = a.dbias + (long long)h * a.T * a.Sp;
    float* o1 = a.drel1d + ((long long)h * DB_NPARTS + part) * (2 * Lt - 1);
    // 64 diagonals per pass, the rows of a diagonal dealt to four threads (a long prompt -- 215 text tokens at 150 classes --
    // made one thread per diagonal a chain of 54 dependent round trips); the four sums meet in LDS in a fixed order
    const int dl = tid & 63, rc = tid >> 6;
    for (int d0 = sub; d0 < 2 * Lt - 1; d0 += DB_NSUB * 64) {
      const int d = d0 + DB_NSUB * dl;
      float t = 0.f;
      if (d < 2 * Lt - 1) {
        const int off = d - (Lt - 1);          // i - j
        const int lo = off > 0 ? off : 0, hi = off < 0 ? Lt + off : Lt;
        const int first = lo + ((part - lo) % DB_NPARTS + DB_NPARTS) % DB_NPARTS;
#pragma unroll 4
        for (int ti = first + rc * DB_NPARTS; ti < hi; ti += 4 * DB_NPARTS) {
          const long long o = (long long)(P + ti) * a.Sp + P + ti - off;
          t += bf2f(hb[o]);
          if (a.ng > 1) t += bf2f(hb[a.gs + o]);
        }
      }
      __syncthreads();
      sf[rc * 64 + dl] = t;
      __syncthreads();
      if (rc == 0 && d < 2 * Lt - 1) {
        const float tt = (sf[dl] + sf[64 + dl]) + (sf[128 + dl] + sf[192 + dl]);
        o1[d] = (a.tab_accumulate ? o1[d] : 0.f) + tt;
      }
    }
    return;
  }
  {
    // ---- (head, part): the two scalar bias entries (grid row x tail column, tail row x grid column) and the tail x tail
    // Toeplitz table, over the rows i = part (mod DB_NPARTS); fixed summation order
    const int h = blk / DB_NPARTS, part = blk % DB_NPARTS, P = a.P, Lt = a.Lt;
    const bf16_t* hb = a.dbias + (long long)h * a.T * a.Sp;
    auto dB = [&](int i, int j) {
      float t = bf2f(hb[(long long)i * a.Sp + j]);
      if (a.ng > 1) t += bf2f(hb[a.gs + (long long)i * a.Sp + j]);
      return t;
    };
    // grid rows x tail columns: (row, 16-byte chunk) pairs dealt to the threads chunk-fastest -- neighbouring lanes read
    // neighbouring chunks of one row (a thread per row read 2 KB-strided pieces, 27 dependent steps at 215 text tokens) --
    // eight loads in flight per thread; the columns past the last whole chunk one element at a time (P is a multiple of 8)
    float t0 = 0.f, t1 = 0.f;
    {
      const int nrow = (P - part + DB_NPARTS - 1) / DB_NPARTS, nch = Lt >> 3, nit = nrow * nch;
      float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int it0 = tid; it0 < nit; it0 += 4 * 256) {
        uint4 v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int it = it0 + u * 256, r = it / nch, ch = it - r * nch;
#pragma unroll
          for (int g = 0; g < 2; ++g)
            v[u][g] = (it < nit && g < a.ng) ? *reinterpret_cast<const uint4*>(hb + g * a.gs + (long long)(part + DB_NPARTS * r) * a.Sp + P + ch * 8)
                                             : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { add8(t, v[u][0]); add8(t, v[u][1]); }
      }
      float tail = 0.f;
      const int nt = Lt & 7;
      for (int it = tid; it < nrow * nt; it += 256) {
        const int r = it / nt, e = it - r * nt;
        tail += dB(part + DB_NPARTS * r, P + (Lt & ~7) + e);
      }
      t0 = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])) + tail;
    }
    // tail rows x grid columns: coalesced 16-byte chunks along the row
    // (four rows' loads in flight together: a long tail -- 239 rows at 640 x 640 / 171 classes -- made this the kernel's
    // longest block by far, 60 dependent round trips)
    // (column chunks x two interleaved row sets, so that a 32-wide grid's 128 chunks still occupy all 256 threads)
    const int nchunk = P >> 3, nrs = nchunk <= 128 ? 2 : 1;
    for (int cc = tid; cc < nchunk * nrs; cc += 256) {
      const int c = cc % nchunk, rs = cc / nchunk;
      float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int ti = part + rs * DB_NPARTS;
      const int tstep = nrs * DB_NPARTS;
      for (; ti + 3 * tstep < Lt; ti += 4 * tstep) {
        uint4 v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int g = 0; g < 2; ++g)
            v[u][g] = g < a.ng ? *reinterpret_cast<const uint4*>(hb + g * a.gs + (long long)(P + ti + u * tstep) * a.Sp + c * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) { add8(t, v[u][0]); add8(t, v[u][1]); }
      }
      for (; ti < Lt; ti += tstep)
        for (int g = 0; g < a.ng; ++g) add8(t, *reinterpret_cast<const uint4*>(hb + g * a.gs + (long long)(P + ti) * a.Sp + c * 8));
      t1 += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    sf[tid] = t0; sf[256 + tid] = t1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) { sf[tid] += sf[tid + o]; sf[256 + tid] += sf[256 + tid + o]; }
      __syncthreads();
    }
    float* ox = a.drelx + ((long long)h * DB_NPARTS + part) * 2;
    if (tid == 0) { ox[0] = (a.tab_accumulate ? ox[0] : 0.f) + sf[0]; ox[1] = (a.tab_accumulate ? ox[1] : 0.f) + sf[256]; }
    return;
  }
}

}  // namespace

extern "C" int ifseg_attn_dense_bias(const void* pos_q, const void* pos_k, int ldpq, int ldpk, int H, int T, int S,
                                     int rel_mode, int P, const int* gcode, int code_bias, int n2d, const float* rel2d,
                                     const float* rel1d, const float* relx, int causal, void* D, int Sp, int Tp,
                                     void* stream) {
  (void)hipGetLastError();
  if (!D || H <= 0 || T <= 0 || S <= 0 || (Sp & 31) || (Tp & 31) || Sp < S || Tp < T) return IFSEG_ERR_BAD_ARG;
  if ((pos_q == nullptr) != (pos_k == nullptr) || ((ldpq | ldpk) & 7)) return IFSEG_ERR_BAD_ARG;
  if (rel_mode && (!gcode || !rel2d || !rel1d || !relx)) return IFSEG_ERR_BAD_ARG;
  if ((rel_mode || causal) && (P > T || P > S || P < 0)) return IFSEG_ERR_BAD_SHAPE;
  DenseArgs a{};
  a.pq = (const bf16_t*)pos_q; a.pk = (const bf16_t*)pos_k; a.ldpq = ldpq; a.ldpk = ldpk;
  a.H = H; a.T = T; a.S = S; a.Sp = Sp; a.Tp = Tp;
  a.rel_mode = rel_mode; a.P = (rel_mode || causal) ? P : S; a.code_bias = code_bias; a.n2d = rel_mode ? n2d : 0; a.Lt = rel_mode ? T - P : 0; a.causal = causal;
  a.gcode = gcode; a.rel2d = rel2d; a.rel1d = rel1d; a.relx = relx; a.D = (bf16_t*)D;
  if ((size_t)D & 15) return IFSEG_ERR_BAD_ARG;
  const size_t lds = rel_mode ? ((((size_t)n2d + 3) & ~(size_t)3) + (((size_t)2 * a.Lt + 2) & ~(size_t)3) + (size_t)P) * 4 : 16;
  if (lds > 160 * 1024) return IFSEG_ERR_BAD_SHAPE;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_dense_bias_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(attn_dense_bias_kernel, dim3((unsigned)(H * (Tp / 32))), dim3(512), lds, (hipStream_t)stream, a);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_attn_bwd_bi(const ifseg_attn_bi_args* x, void* stream) {
  (void)hipGetLastError();
  if (!x || x->B <= 0 || x->H <= 0 || x->T <= 0 || x->S <= 0) return IFSEG_ERR_BAD_ARG;
  if ((x->Sp & 31) || (x->Tp & 31) || x->Sp < x->S || x->Tp < x->T || !x->D) return IFSEG_ERR_BAD_ARG;
  BiArgs a{};
  a.q = (const bf16_t*)x->q; a.k = (const bf16_t*)x->k; a.v = (const bf16_t*)x->v; a.dO = (const bf16_t*)x->dout;
  a.lse = x->lse; a.delta = x->delta; a.D = (const bf16_t*)x->D; a.gain = (const float*)x->gain;
  a.dq = (bf16_t*)x->dq; a.dk = (bf16_t*)x->dk; a.dv = (bf16_t*)x->dv; a.dbias = (bf16_t*)x->dbias; a.dgain_rows = x->dgain_rows;
  a.B = x->B; a.H = x->H; a.T = x->T; a.S = x->S; a.Sp = x->Sp; a.Tp = x->Tp;
  a.q_bs = x->q_bs; a.k_bs = x->k_bs; a.v_bs = x->v_bs; a.do_bs = x->do_bs; a.dq_bs = x->dq_bs; a.dk_bs = x->dk_bs; a.dv_bs = x->dv_bs;
  a.dbias_gs = (long long)x->H * x->T * x->Sp;
  a.ldq = x->ldq; a.ldk = x->ldk; a.ldv = x->ldv; a.lddo = x->lddo; a.lddq = x->lddq; a.lddk = x->lddk; a.lddv = x->lddv;
  a.causal = x->causal; a.P = x->causal ? x->P : x->S; a.dq_scale = x->dq_scale; a.kv_len = x->kv_len;
  a.drop_p = x->drop_p; a.drop_seed = x->drop_seed; a.drop_seed_add = x->drop_seed_add;
  if (!(a.drop_p >= 0.f && a.drop_p < 1.f)) return IFSEG_ERR_BAD_ARG;
  const bool dropping = a.drop_p > 0.f;
  if ((a.ldq | a.ldk | a.ldv | a.lddo | a.lddq | a.lddk | a.lddv) & 7) return IFSEG_ERR_BAD_SHAPE;
  if (((size_t)a.q | (size_t)a.k | (size_t)a.v | (size_t)a.dO | (size_t)a.dq | (size_t)a.dk | (size_t)a.dv | (size_t)a.dbias) & 15)
    return IFSEG_ERR_BAD_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.do_bs | a.dq_bs | a.dk_bs | a.dv_bs) & 7) return IFSEG_ERR_BAD_SHAPE;
  if (a.causal && ((a.P & 63) || a.P > a.T || a.P > a.S)) return IFSEG_ERR_BAD_SHAPE;
  {   // rows are addressed by 32-bit byte offsets from a per-(batch, head) or per-head base
    long long ldmax = a.ldq;
    for (long long l : {(long long)a.lddo, (long long)a.ldk, (long long)a.ldv}) ldmax = l > ldmax ? l : ldmax;
    const long long rows = a.T > a.S ? a.T : a.S;
    if (rows * ldmax * 2 >= (1ll << 31) || (long long)a.Tp * a.Sp * 2 >= (1ll << 31))
      return IFSEG_ERR_BAD_SHAPE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nbg = (a.B + 3) / 4;
  const int ph = x->phases ? x->phases : (IFSEG_ATTN_BWD_DKV | IFSEG_ATTN_BWD_DQ);
  if (ph & IFSEG_ATTN_BWD_DKV) {
    if (!a.dk || !a.dv) return IFSEG_ERR_BAD_ARG;
    auto kern = dropping ? attn_bi_dkv_drop_kernel : attn_bi_dkv_kernel;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
    ifseg_prof_begin(IFSEG_K_ATTN_DKV, s, 6.0 * 64 * (double)a.T * a.S * a.B * a.H, 0);
    hipLaunchKernelGGL(kern, dim3(((a.S + 63) / 64) * a.H * nbg), dim3(512), LDS_DKV, s, a);
    ifseg_prof_end(IFSEG_K_ATTN_DKV, s);
  }
  if (ph & IFSEG_ATTN_BWD_DQ) {
    if (!a.dq || !a.dbias) return IFSEG_ERR_BAD_ARG;
    auto kern = dropping ? attn_bi_dq_drop_kernel : attn_bi_dq_kernel;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
    ifseg_prof_begin(IFSEG_K_ATTN_DQ, s, 2.0 * 64 * (double)a.T * a.S * a.B * a.H, 0);
    hipLaunchKernelGGL(kern, dim3(((a.T + 63) / 64) * a.H * nbg), dim3(512), LDS_DQ, s, a);
    ifseg_prof_end(IFSEG_K_ATTN_DQ, s);
  }
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_attn_fwd_bi(const ifseg_attn_bi_args* x, void* stream) {
  (void)hipGetLastError();
  if (!x || x->B <= 0 || x->H <= 0 || x->T <= 0 || x->S <= 0 || !x->q || !x->k || !x->v || !x->out || !x->lse || !x->D) return IFSEG_ERR_BAD_ARG;
  if ((x->Sp & 31) || x->Sp < x->S || (x->Tp & 31) || x->Tp < x->T) return IFSEG_ERR_BAD_ARG;
  BiArgs a{};
  a.q = (const bf16_t*)x->q; a.k = (const bf16_t*)x->k; a.v = (const bf16_t*)x->v; a.D = (const bf16_t*)x->D; a.gain = (const float*)x->gain;
  a.out = (bf16_t*)x->out; a.lse_out = const_cast<float*>(x->lse); a.o_bs = x->out_bs; a.ldo = x->ldout;
  a.B = x->B; a.H = x->H; a.T = x->T; a.S = x->S; a.Sp = x->Sp; a.Tp = x->Tp;
  a.q_bs = x->q_bs; a.k_bs = x->k_bs; a.v_bs = x->v_bs; a.ldq = x->ldq; a.ldk = x->ldk; a.ldv = x->ldv;
  a.causal = x->causal; a.P = x->causal ? x->P : x->S; a.kv_len = x->kv_len;
  a.drop_p = x->drop_p; a.drop_seed = x->drop_seed; a.drop_seed_add = x->drop_seed_add;
  if (!(a.drop_p >= 0.f && a.drop_p < 1.f)) return IFSEG_ERR_BAD_ARG;
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return IFSEG_ERR_BAD_SHAPE;
  if (((size_t)a.q | (size_t)a.k | (size_t)a.v | (size_t)a.out) & 15) return IFSEG_ERR_BAD_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs) & 7) return IFSEG_ERR_BAD_SHAPE;
  if (a.causal && ((a.P & 63) || a.P > a.T || a.P > a.S)) return IFSEG_ERR_BAD_SHAPE;
  {
    const long long ldmax = a.ldk > a.ldv ? a.ldk : a.ldv;
    if ((long long)a.S * ldmax * 2 >= (1ll << 31) || (long long)a.Tp * a.Sp * 2 >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int lds = 2 * STG_DQ;
  auto kern = a.drop_p > 0.f ? attn_bi_fwd_drop_kernel : attn_bi_fwd_kernel;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  ifseg_prof_begin(IFSEG_K_ATTN_FWD, s, 4.0 * 64 * (double)a.T * a.S * a.B * a.H, 0);
  hipLaunchKernelGGL(kern, dim3(((a.T + 63) / 64) * a.H * ((a.B + 3) / 4)), dim3(512), lds, s, a);
  ifseg_prof_end(IFSEG_K_ATTN_FWD, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_attn_dbias_grads(const ifseg_attn_dbias_args* x, void* stream) {
  (void)hipGetLastError();
  if (!x || !x->dbias || x->ng <= 0 || x->H <= 0 || x->T <= 0 || x->S <= 0 || (x->Sp & 31) || x->Sp < x->S)
    return IFSEG_ERR_BAD_ARG;
  DbArgs a{};
  a.dbias = (const bf16_t*)x->dbias; a.gs = (long long)x->H * x->T * x->Sp; a.ng = x->ng;
  a.H = x->H; a.T = x->T; a.S = x->S; a.Sp = x->Sp; a.C = x->C;
  a.pq = (const bf16_t*)x->pos_q; a.pk = (const bf16_t*)x->pos_k; a.ldpq = x->ldpq; a.ldpk = x->ldpk;
  a.dpq = x->dpos_q_acc; a.dpk = x->dpos_k_acc; a.accumulate = x->accumulate_pos; a.dpq_scale = x->dpq_scale;
  a.P = x->P; a.gh = x->grid_h; a.gw = x->grid_w; a.Lt = x->T - x->P; a.causal = x->causal;
  if (a.causal && (a.P <= 0 || (a.P & 31))) return IFSEG_ERR_BAD_SHAPE;
  a.drel2d = x->drel2d; a.drel1d = x->drel1d; a.drelx = x->drelx;
  const bool pos = a.pq != nullptr;
  if (pos && (!a.pk || !a.dpq || !a.dpk || ((a.ldpq | a.ldpk) & 7) || (a.C & 3) || a.C < a.H * 64)) return IFSEG_ERR_BAD_ARG;
  const bool rel = a.drel2d != nullptr;
  if (rel) {
    if (!a.drel1d || !a.drelx || a.gh <= 0 || a.gw <= 0 || a.gw > 64 || (a.gw & 7) || a.gh * a.gw != a.P || a.P > a.T || a.P > a.S || a.T != a.S)
      return IFSEG_ERR_BAD_SHAPE;
  }
  a.nb_q = pos ? a.H * ((a.T + 31) / 32) : 0;
  a.nb_k = pos ? a.H * (a.Sp / 32) : 0;
  a.nb_2d = rel ? a.H * (2 * a.gh - 1) * DB_NPARTS : 0;
  // the kernels take the slabs (groups of four batch elements) two at a time: batches of more than eight per GPU run them
  // once per pair, later pairs adding to the first pair's results in launch order (deterministic)
  const int ng_all = x->ng;
  for (int g0 = 0; g0 < ng_all; g0 += 2) {
    a.dbias = (const bf16_t*)x->dbias + (long long)g0 * a.gs;
    a.ng = ng_all - g0 < 2 ? ng_all - g0 : 2;
    if (g0) { a.accumulate = 1; a.tab_accumulate = 1; }
    if (pos) hipLaunchKernelGGL(attn_dbias_grads_kernel, dim3(a.nb_q + a.nb_k), dim3(256), 0, (hipStream_t)stream, a);
    if (rel) hipLaunchKernelGGL(attn_dbias_tables_kernel, dim3(a.nb_2d + a.H * DB_NPARTS * (1 + DB_NSUB)), dim3(256), 0, (hipStream_t)stream, a);
  }
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_attn_dbias_nparts(void) { return DB_NPARTS; }

extern "C" int ifseg_attn_dropout_mask(unsigned char* keep, int B, int H, int T, int S, float p, unsigned long long seed,
                                       const unsigned long long* seed_add, void* stream) {
  (void)hipGetLastError();
  if (!keep || B <= 0 || H <= 0 || T <= 0 || S <= 0 || !(p >= 0.f && p < 1.f) || (long long)B * H * T >= (1ll << 32)) return IFSEG_ERR_BAD_ARG;
  const long long n = (long long)B * H * T * S;
  hipLaunchKernelGGL(attn_drop_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keep, B, H, T, S, p, seed, seed_add);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
