// HBM-bound row kernels for the SegOFA blocks (gfx950): LayerNorm forward/backward
// with fused GELU input and residual output, partial-sum reductions, bias-gradient
// column sums, embedding gathers.  16-byte vector accesses, one wave per row,
// wave shuffles for the row statistics; no global atomics (deterministic).
//
// Reference ops replaced: fairseq LayerNorm (modules/layer_norm.py:30-35, eps 1e-5)
// at every *_layer_norm / attn_ln / ffn_layernorm / layernorm_embedding site of
// unify_transformer_layer.py:256-292,463-568, encoder_module.py:405-424,757-760,829,
// decoder_module.py:342-346,575-576,668; GELU in fp32 (modules/gelu.py:24-25);
// residual_connection (unify_transformer_layer.py:196); their autograd.
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {

struct RowMap {  // logical row r -> element offset  (r / rpb) * bs + (r % rpb) * ld
  int rpb; long long bs; int ld;
  __device__ __forceinline__ long long off(int r) const {
    return rpb > 0 ? (long long)(r / rpb) * bs + (long long)(r % rpb) * ld : (long long)r * ld;
  }
};
// host side: a [B, rpb, C] view whose batches follow each other without a gap (bs == rpb * ld: every full-width activation
// tensor of the engine) is the plain map r -> r * ld.  The kernels branch on rpb (uniform): the plain map skips an integer
// division by a run-time value + 64-bit multiply-adds -- ~28 dependent VALU instructions per operand and ROW in front of the row's
// first load (4-5 operands per LayerNorm launch).  Round 6.
static inline RowMap row_map(int rpb, long long bs, int ld) {
  return (rpb > 0 && bs == (long long)rpb * ld) ? RowMap{0, 0, ld} : RowMap{rpb, bs, ld};
}

__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.f + erf_as(x, __expf(-0.5f * x * x)));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// ---- dropout + DropPath (training-time stochastic regularisers) -------------------------------
// keep(i) ~ Bernoulli(1 - p) from a counter-based hash of (seed, 8-element chunk index): the backward
// re-generates the same mask, nothing is stored.  Reference: FairseqDropout (fairseq_dropout.py:23-27)
// after the embedding LayerNorms, attn_ln / cross_attn_ln and fc2, and drop_path
// (unify_transformer_layer.py:19-35) inside residual_connection (:196).  The RNG stream necessarily
// differs from torch's.  The same mask is applied by the stand-alone kernel (ifseg_dropout) and by the
// fused epilogue of ln_fwd / prologue of ln_bwd.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct DropArgs {   // on == 0: identity
  int on; float p; unsigned long long seed; const float* dpscale; int rows_per_batch;
  const unsigned long long* seed_add;   // device word added to `seed` (the per-update part: a captured step replays with new masks)
};
// f[0..8) *= keep * scale for chunk c8 (= row * C/8 + chunk) of logical row `row`
__device__ __forceinline__ void drop8(float* f, const DropArgs& d, long long c8, int row) {
  const float sc = (d.dpscale ? d.dpscale[row / d.rows_per_batch] : 1.f) * (d.p > 0.f ? 1.f / (1.f - d.p) : 1.f);
  const unsigned thr = (unsigned)(d.p * 65536.f);
  const unsigned long long sd = d.seed + (d.seed_add ? *d.seed_add : 0ull);
  const unsigned long long r0 = splitmix64(sd + 2ull * (unsigned long long)c8), r1 = splitmix64(sd + 2ull * (unsigned long long)c8 + 1ull);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned bits = (unsigned)(((e < 4 ? r0 : r1) >> (16 * (e & 3))) & 0xFFFFu);
    f[e] = bits >= thr ? f[e] * sc : 0.f;
  }
}

// LayerNorm gains / biases: bf16 (the model's arena) or fp32 (the fp32 master copy: rounding gamma / beta of ~50
// LayerNorms to bf16 alone costs 0.9e-2 of logits rel-L2 on SegOFA-Base, tools/err_budget2.py); wave-uniform choice
__device__ __forceinline__ void ldp8(const bf16_t* p, int c, int pf32, float* o) {
  if (pf32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + c * 8);
    const float4 a = q[0], b = q[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    unpack8(*reinterpret_cast<const uint4*>(p + c * 8), o);
  }
}

// y = [resid +] drop(LN(act(x)) * gamma + beta) ; one wave per row, NCH 8-element chunks per lane
template <int NCH, bool GELU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta,
                                                     const bf16_t* resid, bf16_t* y, float* mean, float* rstd,
                                                     int rows, int C, float eps, RowMap mx, RowMap my, RowMap mr,
                                                     DropArgs drop, const bf16_t* gamma2, const bf16_t* beta2, bf16_t* y2,
                                                     float* mean2, float* rstd2, RowMap my2, int pf32) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C >> 3;
  const bf16_t* xp = x + mx.off(row);
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      uint4 u = *reinterpret_cast<const uint4*>(xp + c * 8);
      unpack8(u, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { if (GELU) v[i][e] = gelu_f(v[i][e]); s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  // gamma == nullptr: the first stage is the identity (y = [resid +] drop(x)), only the second LayerNorm runs
  float mu = 0.f, rs = 1.f;
  if (gamma) {
    mu = warp_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    rs = rsqrtf(warp_sum(q) / C + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
  }
  bf16_t* yp = y + my.off(row);
  const bf16_t* rp = resid ? resid + mr.off(row) : nullptr;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      float o[8];
      if (gamma) {
        float g[8], b[8];
        ldp8(gamma, c, pf32, g);
        ldp8(beta, c, pf32, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[i][e];
      }
      if (drop.on) drop8(o, drop, (long long)row * nch + c, row);
      if (rp) {
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(rp + c * 8), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r[e];
      }
      const uint4 packed = pack8(o);
      *reinterpret_cast<uint4*>(yp + c * 8) = packed;
      if (y2) unpack8(packed, v[i]);        // the second LayerNorm sees y exactly as it is stored (bf16)
    }
  }
  // ---- optional second LayerNorm of the row just produced: y2 = LN2(y)  (the pre-LN of the next block, fused so
  // the residual stream is not read back: attn_ln -> +x -> final_layer_norm, unify_transformer_layer.py:256-292)
  if (y2) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 += v[i][e];
      }
    const float mu2 = warp_sum(s2) / C;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu2; q2 += d * d; }
      }
    const float rs2 = rsqrtf(warp_sum(q2) / C + eps);
    if (lane == 0 && mean2) { mean2[row] = mu2; rstd2[row] = rs2; }
    bf16_t* y2p = y2 + my2.off(row);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float g[8], b[8], o[8];
        ldp8(gamma2, c, pf32, g);
        ldp8(beta2, c, pf32, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu2) * rs2 * g[e] + b[e];
        *reinterpret_cast<uint4*>(y2p + c * 8) = pack8(o);
      }
    }
  }
}

// backward: dx = [dx_add +] act'(x) * rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma
// dgamma/dbeta: per-block partial sums over the rows the block visited -> [nblk][C]
// WPR = waves per row: 1 (narrow rows: a wave owns a row, four rows per block in flight) or 4 (wide rows: the
// block owns a row, a thread keeps NCH <= 2 chunks so the three per-column accumulators stay in registers).
// The next row's x / dy are fetched (as packed bf16) while the current row is reduced.
template <int NCH, bool GELU, int WPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* gamma,
                                                     const float* mean, const float* rstd, const bf16_t* dx_add,
                                                     bf16_t* dx, float* dgamma_part, float* dbeta_part, int rows, int C,
                                                     RowMap mdy, RowMap mx, RowMap mdx, RowMap madd, DropArgs drop, int pf32) {
  __shared__ float red[(WPR == 1) ? 4 * (64 * 8 + 8) : 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  const int tl = (WPR == 1) ? lane : threadIdx.x;      // chunk lane of this thread inside its row
  constexpr int TS = 64 * WPR;                          // chunk stride
  float gam[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tl + i * TS;
    if (c < nch) ldp8(gamma, c, pf32, gam[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; if (c >= nch) gam[i][e] = 0.f; }
  }
  const int row0 = (WPR == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
  const int rstep = (WPR == 1) ? gridDim.x * 4 : gridDim.x;
  uint4 rx[NCH], rd[NCH];
  auto fetch = [&](int row) {
    const bf16_t* xp = x + mx.off(row);
    const bf16_t* dyp = dy + mdy.off(row);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tl + i * TS;
      if (c < nch) {
        rx[i] = *reinterpret_cast<const uint4*>(xp + c * 8);
        rd[i] = *reinterpret_cast<const uint4*>(dyp + c * 8);
      }
    }
  };
  if (row0 < rows) fetch(row0);
  int it = 0;
  for (int row = row0; row < rows; row += rstep, ++it) {
    const float mu = mean[row], rs = rstd[row];
    float xr[NCH][8], gv[NCH][8], da[GELU ? NCH : 1][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tl + i * TS;
      if (c < nch) {
        float d[8];
        unpack8(rx[i], xr[i]);
        unpack8(rd[i], d);
        if (drop.on) drop8(d, drop, (long long)row * nch + c, row);     // adjoint of the forward's fused dropout
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = xr[i][e];
          if (GELU) {
            const float xv = xr[i][e], ex = __expf(-0.5f * xv * xv), er = erf_as(xv, ex);
            a = 0.5f * xv * (1.f + er);
            da[i][e] = 0.5f * (1.f + er) + xv * 0.39894228040143268f * ex;
          }
          const float xh = (a - mu) * rs;
          const float g = d[e] * gam[i][e];
          dg[i][e] += d[e] * xh; db[i][e] += d[e];
          s1 += g; s2 += g * xh;
          gv[i][e] = g;
          xr[i][e] = xh;
        }
      }
    }
    if (row + rstep < rows) fetch(row + rstep);      // in flight under the reduction and the store below
    s1 = warp_sum(s1); s2 = warp_sum(s2);
    if (WPR > 1) {
      float* slot = red + (it & 1) * 8;               // double-buffered: one barrier per row
      if (lane == 0) { slot[wave * 2] = s1; slot[wave * 2 + 1] = s2; }
      __syncthreads();
      s1 = (slot[0] + slot[2]) + (slot[4] + slot[6]);
      s2 = (slot[1] + slot[3]) + (slot[5] + slot[7]);
    }
    s1 /= C; s2 /= C;
    bf16_t* dxp = dx + mdx.off(row);
    const bf16_t* ap = dx_add ? dx_add + madd.off(row) : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tl + i * TS;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = xr[i][e], dact = GELU ? da[i][e] : 1.f;
          o[e] = rs * (gv[i][e] - s1 - xh * s2) * dact;
        }
        if (ap) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(ap + c * 8), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
        *reinterpret_cast<uint4*>(dxp + c * 8) = pack8(o);
      }
    }
  }
  if (!dgamma_part) return;
  if (WPR > 1) {
    // every thread owns its columns: the block's partial sums go straight out
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tl + i * TS;
      if (c < nch) {
        float* gp = dgamma_part + (long long)blockIdx.x * C + c * 8;
        float* bp = dbeta_part + (long long)blockIdx.x * C + c * 8;
        *reinterpret_cast<float4*>(gp) = make_float4(dg[i][0], dg[i][1], dg[i][2], dg[i][3]);
        *reinterpret_cast<float4*>(gp + 4) = make_float4(dg[i][4], dg[i][5], dg[i][6], dg[i][7]);
        *reinterpret_cast<float4*>(bp) = make_float4(db[i][0], db[i][1], db[i][2], db[i][3]);
        *reinterpret_cast<float4*>(bp + 4) = make_float4(db[i][4], db[i][5], db[i][6], db[i][7]);
      }
    }
  } else {
    // cross-wave reduction of the four rows' partials (chunk by chunk through LDS)
    float (*r4)[64 * 8 + 8] = reinterpret_cast<float (*)[64 * 8 + 8]>(red);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) r4[wave][lane * 8 + e] = pass ? db[i][e] : dg[i][e];
        __syncthreads();
        for (int t = threadIdx.x; t < 512; t += 256) {
          const int c = (t >> 3) + i * 64;
          if (c < nch) {
            const float sum = r4[0][t] + r4[1][t] + r4[2][t] + r4[3][t];
            (pass ? dbeta_part : dgamma_part)[(long long)blockIdx.x * C + c * 8 + (t & 7)] = sum;
          }
        }
      }
    }
  }
}

// ln_bwd, narrow rows (C <= 1024, one wave per row, no GELU), register-lean (round 5): the row's dy / x stay PACKED (bf16)
// across the two row reductions and the second pass re-derives g and xhat from them with the first pass's statements -- g by an
// explicitly rounded multiply -- instead of holding two fp32 copies of the row; no software prefetch.  118 VGPRs instead of
// 146 (4 waves per SIMD), 11-13 % faster alone (tools/ln_bench.py).  Against ln_bwd_kernel<2, false, 1> the per-block partial
// sums are bit-identical and dx differs in ~1e-5 of its elements by one bf16 ulp (tools/ln_hash.py: the compiler fuses the
// row sums' multiply-adds differently in the two kernels).  DX2: the second output of ifseg_ln_bwd_drop,
// dx2 = drop2(dx as stored in bf16) -- the same instantiation as the plain call, so the pair stays bit-identical to
// ifseg_ln_bwd followed by ifseg_dropout.  (ln_bwd_kernel remains for the GELU / wide rows.)
template <int NCH, bool DX2>
__global__ __launch_bounds__(256) void ln_bwd_lean_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* gamma,
                                                          const float* mean, const float* rstd, const bf16_t* dx_add,
                                                          bf16_t* dx, float* dgamma_part, float* dbeta_part, int rows, int C,
                                                          RowMap mdy, RowMap mx, RowMap mdx, RowMap madd, DropArgs drop, int pf32,
                                                          bf16_t* dx2, RowMap mdx2, DropArgs drop2) {
  __shared__ float red[4 * (64 * 8 + 8)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  float gam[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) ldp8(gamma, c, pf32, gam[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; if (c >= nch) gam[i][e] = 0.f; }
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const bf16_t* xp = x + mx.off(row);
    const bf16_t* dyp = dy + mdy.off(row);
    uint4 rx[NCH], rd[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        rx[i] = *reinterpret_cast<const uint4*>(xp + c * 8);
        rd[i] = *reinterpret_cast<const uint4*>(dyp + c * 8);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float xr[8], d[8];
        unpack8(rx[i], xr);
        unpack8(rd[i], d);
        if (drop.on) drop8(d, drop, (long long)row * nch + c, row);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xr[e] - mu) * rs;
          const float g = d[e] * gam[i][e];
          dg[i][e] += d[e] * xh; db[i][e] += d[e];
          s1 += g; s2 += g * xh;
        }
      }
    }
    s1 = warp_sum(s1); s2 = warp_sum(s2);
    s1 /= C; s2 /= C;
    // (opaque to the optimiser: the second pass must RE-derive xhat and g from the packed row instead of keeping the first
    // pass's 32 fp32 values alive across the reductions)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      asm volatile("" : "+v"(rx[i].x), "+v"(rx[i].y), "+v"(rx[i].z), "+v"(rx[i].w), "+v"(rd[i].x), "+v"(rd[i].y), "+v"(rd[i].z), "+v"(rd[i].w));
    bf16_t* dxp = dx + mdx.off(row);
    const bf16_t* ap = dx_add ? dx_add + madd.off(row) : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float xr[8], d[8], o[8];
        unpack8(rx[i], xr);
        unpack8(rd[i], d);
        if (drop.on) drop8(d, drop, (long long)row * nch + c, row);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xr[e] - mu) * rs;
          const float g = __fmul_rn(d[e], gam[i][e]);           // (rounded as the first pass's g was)
          o[e] = rs * (g - s1 - xh * s2) * 1.f;
        }
        if (ap) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(ap + c * 8), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
        const uint4 pk = pack8(o);
        *reinterpret_cast<uint4*>(dxp + c * 8) = pk;
        if (DX2) {
          float d2[8];
          unpack8(pk, d2);                               // the adjoint sees dx as stored
          if (drop2.on) drop8(d2, drop2, (long long)row * nch + c, row);
          *reinterpret_cast<uint4*>(dx2 + mdx2.off(row) + c * 8) = pack8(d2);
        }
      }
    }
  }
  if (!dgamma_part) return;
  float (*r4)[64 * 8 + 8] = reinterpret_cast<float (*)[64 * 8 + 8]>(red);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) r4[wave][lane * 8 + e] = pass ? db[i][e] : dg[i][e];
      __syncthreads();
      for (int t = threadIdx.x; t < 512; t += 256) {
        const int c = (t >> 3) + i * 64;
        if (c < nch) {
          const float sum = r4[0][t] + r4[1][t] + r4[2][t] + r4[3][t];
          (pass ? dbeta_part : dgamma_part)[(long long)blockIdx.x * C + c * 8 + (t & 7)] = sum;
        }
      }
    }
  }
}

// out[o][i] (+)= sum_p in[o][p][i]
template <bool OUT_BF16>
__global__ void reduce_parts_kernel(const float* in, void* out, int outer, int parts, long long n, int accumulate,
                                    float scale) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)outer * n) return;
  const long long o = gid / n, i = gid % n;
  const float* p = in + o * parts * n + i;
  float s = 0.f;
  for (int k = 0; k < parts; ++k) s += p[(long long)k * n];
  s *= scale;
  if (OUT_BF16) {
    bf16_t* op = reinterpret_cast<bf16_t*>(out) + gid;
    if (accumulate) s += bf2f(*op);
    *op = f2bf(s);
  } else {
    float* op = reinterpret_cast<float*>(out) + gid;
    if (accumulate) s += *op;
    *op = s;
  }
}

// same reduction for many parts / few columns: 32 columns x 8 part-groups per block
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void reduce_parts2d_kernel(const float* in, void* out, int outer, int parts,
                                                             long long n, int accumulate, float scale) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long nchunk = (n + 31) / 32;
  const long long o = blockIdx.x / nchunk, i = (blockIdx.x % nchunk) * 32 + c;
  float s = 0.f;
  if (i < n) {
    const float* p = in + o * parts * n + i;
#pragma unroll 8
    for (int k = g; k < parts; k += 8) s += p[(long long)k * n];
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    t *= scale;
    const long long gid = o * n + i;
    if (OUT_BF16) {
      bf16_t* op = reinterpret_cast<bf16_t*>(out) + gid;
      if (accumulate) t += bf2f(*op);
      *op = f2bf(t);
    } else {
      float* op = reinterpret_cast<float*>(out) + gid;
      if (accumulate) t += *op;
      *op = t;
    }
  }
}

// the same for MANY parts and few columns (the loss kernel's 8192 per-tile statistics rows of 2 + 3 nseg floats): one block per
// output element, a thread sums every 256th part (its loads independent, in flight together), fixed-order tree over the block.
// (The 32-column kernel above gave each thread 1024 dependent strided loads here: 49 us on the main queue between the forward
// and the backward of every step; this one takes ~6.)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void reduce_parts_tall_kernel(const float* in, void* out, int outer, int parts, long long n,
                                                                int accumulate, float scale) {
  __shared__ float red[256];
  const long long o = blockIdx.x / n, i = blockIdx.x % n;
  const float* p = in + o * parts * n + i;
  float s = 0.f;
#pragma unroll 8
  for (int k = threadIdx.x; k < parts; k += 256) s += p[(long long)k * n];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float t = red[0] * scale;
    const long long gid = o * n + i;
    if (OUT_BF16) {
      bf16_t* op = reinterpret_cast<bf16_t*>(out) + gid;
      if (accumulate) t += bf2f(*op);
      *op = f2bf(t);
    } else {
      float* op = reinterpret_cast<float*>(out) + gid;
      if (accumulate) t += *op;
      *op = t;
    }
  }
}

// several reductions in one launch: block ranges by task
struct ReduceTasks { ifseg_reduce_task t[16]; int start[17]; int n; };
__global__ __launch_bounds__(256) void reduce_parts_multi_kernel(ReduceTasks ts) {
  __shared__ float red[8][33];
  int k = 0;
  for (int i = 1; i < ts.n; ++i) k = ((int)blockIdx.x >= ts.start[i]) ? i : k;
  const ifseg_reduce_task& T = ts.t[k];
  const int blk = blockIdx.x - ts.start[k];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long nchunk = (T.n + 31) / 32;
  const long long o = blk / nchunk, i = (blk % nchunk) * 32 + c;
  float s = 0.f;
  if (i < T.n) {
    const float* p = T.in + o * T.parts * T.n + i;
#pragma unroll 8
    for (int q = g; q < T.parts; q += 8) s += p[(long long)q * T.n];
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && i < T.n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][c];
    t *= T.scale;
    const long long gid = o * T.n + i;
    if (T.out_bf16) {
      bf16_t* op = reinterpret_cast<bf16_t*>(T.out) + gid;
      if (T.accumulate) t += bf2f(*op);
      *op = f2bf(t);
    } else {
      float* op = reinterpret_cast<float*>(T.out) + gid;
      if (T.accumulate) t += *op;
      *op = t;
    }
  }
}

// column sums of a bf16 matrix: part[blockIdx.y][n] = sum over the block's row slab
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, float* part, int M, int N, RowMap mx,
                                                     int rows_per_blk) {
  const int c = blockIdx.x * 256 + threadIdx.x;  // 8-column chunk
  if (c * 8 >= N) return;
  const int r0 = blockIdx.y * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0; r < r1; ++r) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(x + mx.off(r) + c * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  float* o = part + (long long)blockIdx.y * N + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = acc[e];
}

// out[r] = table[ids[r]] + add   (ids int64, C % 8 == 0)
__global__ void embed_rows_kernel(const bf16_t* table, const long long* ids, const bf16_t* add, bf16_t* out, int n,
                                  int C, RowMap mo) {
  const int nch = C >> 3;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n * nch) return;
  const int r = (int)(gid / nch), c = (int)(gid % nch);
  float f[8];
  unpack8(*reinterpret_cast<const uint4*>(table + ids[r] * C + c * 8), f);
  if (add) {
    float a[8];
    unpack8(*reinterpret_cast<const uint4*>(add + c * 8), a);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += a[e];
  }
  *reinterpret_cast<uint4*>(out + mo.off(r) + c * 8) = pack8(f);
}

// elementwise helpers -------------------------------------------------------
__global__ void cast_f32_to_bf16_kernel(const float* in, bf16_t* out, long long n, float scale) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(in + i);
    *reinterpret_cast<uint2*>(out + i) = make_uint2(pack2bf(v.x * scale, v.y * scale), pack2bf(v.z * scale, v.w * scale));
  } else {
    for (long long k = i; k < n; ++k) out[k] = f2bf(in[k] * scale);
  }
}
__global__ void add_bf16_kernel(const bf16_t* a, const bf16_t* b, bf16_t* out, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < n) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const uint4*>(a + i), x);
    unpack8(*reinterpret_cast<const uint4*>(b + i), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    *reinterpret_cast<uint4*>(out + i) = pack8(x);
  } else {
    for (long long k = i; k < n; ++k) out[k] = f2bf(bf2f(a[k]) + bf2f(b[k]));
  }
}
// NCHW (fp32 or bf16) -> NHWC bf16 with channel padding to Cp (zeros)
template <typename TIN>
__global__ void nchw_to_nhwc_kernel(const TIN* in, bf16_t* out, int B, int Cc, int H, int W, int Cp) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * Cp;
  if (gid >= total) return;
  const int c = (int)(gid % Cp);
  const long long p = gid / Cp;
  const int xw = (int)(p % W), yh = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
  float v = 0.f;
  if (c < Cc) {
    const long long src = (((long long)b * Cc + c) * H + yh) * W + xw;
    if (sizeof(TIN) == 4) v = reinterpret_cast<const float*>(in)[src];
    else v = bf2f(reinterpret_cast<const bf16_t*>(in)[src]);
  }
  out[gid] = f2bf(v);
}

template <int NCH>
int launch_ln_fwd(int gelu_flags, dim3 g, hipStream_t s, const bf16_t* x, const bf16_t* gamma, const bf16_t* beta,
                  const bf16_t* resid, bf16_t* y, float* mean, float* rstd, int rows, int C, float eps, RowMap mx,
                  RowMap my, RowMap mr, DropArgs dr, const bf16_t* g2 = nullptr, const bf16_t* b2 = nullptr, bf16_t* y2 = nullptr,
                  float* mean2 = nullptr, float* rstd2 = nullptr, RowMap my2 = RowMap{0, 0, 0}) {
  const int pf32 = (gelu_flags & IFSEG_LN_PARAMS_F32) ? 1 : 0;
  if (gelu_flags & IFSEG_LN_GELU) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true>), g, dim3(256), 0, s, x, gamma, beta, resid, y, mean, rstd, rows, C, eps, mx, my, mr, dr, g2, b2, y2, mean2, rstd2, my2, pf32);
  else hipLaunchKernelGGL((ln_fwd_kernel<NCH, false>), g, dim3(256), 0, s, x, gamma, beta, resid, y, mean, rstd, rows, C, eps, mx, my, mr, dr, g2, b2, y2, mean2, rstd2, my2, pf32);
  return 0;
}
template <int NCH, int WPR>
int launch_ln_bwd(int gelu_flags, dim3 g, hipStream_t s, const bf16_t* dy, const bf16_t* x, const bf16_t* gamma,
                  const float* mean, const float* rstd, const bf16_t* add, bf16_t* dx, float* dgp, float* dbp, int rows,
                  int C, RowMap mdy, RowMap mx, RowMap mdx, RowMap madd, DropArgs dr) {
  const int pf32 = (gelu_flags & IFSEG_LN_PARAMS_F32) ? 1 : 0;
  if (gelu_flags & IFSEG_LN_GELU) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, WPR>), g, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, dgp, dbp, rows, C, mdy, mx, mdx, madd, dr, pf32);
  else hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, WPR>), g, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, dgp, dbp, rows, C, mdy, mx, mdx, madd, dr, pf32);
  return 0;
}

}  // namespace

static int ln_fwd_impl(const void* x, const void* gamma, const void* beta, const void* resid, void* y,
                       float* mean, float* rstd, int rows, int C, float eps, int act_gelu, int rpb,
                       long long x_bs, int ldx, long long y_bs, int ldy, long long r_bs, int ldr,
                       const ifseg_drop_args* drop, const void* gamma2, const void* beta2, void* y2, float* mean2,
                       float* rstd2, long long y2_bs, int ldy2, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if ((C & 7) || C > 4096 || (ldx & 7) || (ldy & 7) || (resid && (ldr & 7)) || (y2 && (ldy2 & 7))) return IFSEG_ERR_BAD_SHAPE;
  if (y2 && (!gamma2 || !beta2)) return IFSEG_ERR_BAD_ARG;
  DropArgs dr{};
  if (drop) {
    if (drop->p < 0.f || drop->p >= 1.f || drop->rows_per_batch <= 0) return IFSEG_ERR_BAD_ARG;
    dr = DropArgs{1, drop->p, drop->seed, drop->drop_path_scale, drop->rows_per_batch, drop->seed_add};
  }
  RowMap mx = row_map(rpb, x_bs, ldx), my = row_map(rpb, y_bs, ldy), mr = row_map(rpb, r_bs, ldr), my2 = row_map(rpb, y2_bs, ldy2);
  dim3 g((rows + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t *X = (const bf16_t*)x, *G = (const bf16_t*)gamma, *Bt = (const bf16_t*)beta, *R = (const bf16_t*)resid;
  const bf16_t *G2 = (const bf16_t*)gamma2, *B2 = (const bf16_t*)beta2;
  ifseg_prof_begin(IFSEG_K_LN_FWD, s, 0, (double)rows * C * ((resid ? 6.0 : 4.0) + (y2 ? 2.0 : 0.0)));
  if (C <= 1024) launch_ln_fwd<2>(act_gelu, g, s, X, G, Bt, R, (bf16_t*)y, mean, rstd, rows, C, eps, mx, my, mr, dr, G2, B2, (bf16_t*)y2, mean2, rstd2, my2);
  else if (C <= 3072) launch_ln_fwd<6>(act_gelu, g, s, X, G, Bt, R, (bf16_t*)y, mean, rstd, rows, C, eps, mx, my, mr, dr, G2, B2, (bf16_t*)y2, mean2, rstd2, my2);
  else launch_ln_fwd<8>(act_gelu, g, s, X, G, Bt, R, (bf16_t*)y, mean, rstd, rows, C, eps, mx, my, mr, dr, G2, B2, (bf16_t*)y2, mean2, rstd2, my2);
  ifseg_prof_end(IFSEG_K_LN_FWD, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_ln_fwd(const void* x, const void* gamma, const void* beta, const void* resid, void* y,
                            float* mean, float* rstd, int rows, int C, float eps, int act_gelu, int rpb,
                            long long x_bs, int ldx, long long y_bs, int ldy, long long r_bs, int ldr,
                            const ifseg_drop_args* drop, void* stream) {
  return ln_fwd_impl(x, gamma, beta, resid, y, mean, rstd, rows, C, eps, act_gelu, rpb, x_bs, ldx, y_bs, ldy, r_bs, ldr, drop,
                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int ifseg_ln_fwd_pair(const void* x, const void* gamma, const void* beta, const void* resid, void* y,
                                 float* mean, float* rstd, const void* gamma2, const void* beta2, void* y2, float* mean2,
                                 float* rstd2, int rows, int C, float eps, int flags, int rpb, long long x_bs, int ldx, long long y_bs,
                                 int ldy, long long r_bs, int ldr, long long y2_bs, int ldy2, const ifseg_drop_args* drop,
                                 void* stream) {
  if (!y2) return IFSEG_ERR_BAD_ARG;
  if (!gamma && beta) return IFSEG_ERR_BAD_ARG;
  if (flags & IFSEG_LN_GELU) return IFSEG_ERR_BAD_ARG;
  return ln_fwd_impl(x, gamma, beta, resid, y, mean, rstd, rows, C, eps, flags, rpb, x_bs, ldx, y_bs, ldy, r_bs, ldr, drop, gamma2,
                     beta2, y2, mean2, rstd2, y2_bs, ldy2, stream);
}

extern "C" int ifseg_ln_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                            const void* dx_add, void* dx, float* dgamma_part, float* dbeta_part, int nblocks,
                            int rows, int C, int act_gelu, int rpb, long long dy_bs, int lddy, long long x_bs,
                            int ldx, long long dx_bs, int lddx, long long add_bs, int ldadd,
                            const ifseg_drop_args* drop, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if ((C & 7) || C > 4096 || nblocks <= 0) return IFSEG_ERR_BAD_SHAPE;
  DropArgs dr{};
  if (drop) {
    if (drop->p < 0.f || drop->p >= 1.f || drop->rows_per_batch <= 0) return IFSEG_ERR_BAD_ARG;
    dr = DropArgs{1, drop->p, drop->seed, drop->drop_path_scale, drop->rows_per_batch, drop->seed_add};
  }
  RowMap mdy = row_map(rpb, dy_bs, lddy), mx = row_map(rpb, x_bs, ldx), mdx = row_map(rpb, dx_bs, lddx), madd = row_map(rpb, add_bs, ldadd);
  dim3 g(nblocks);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t *DY = (const bf16_t*)dy, *X = (const bf16_t*)x, *G = (const bf16_t*)gamma, *A = (const bf16_t*)dx_add;
  ifseg_prof_begin(IFSEG_K_LN_BWD, s, 0, (double)rows * C * (dx_add ? 8.0 : 6.0));
  if (C <= 1024 && !(act_gelu & IFSEG_LN_GELU))
    hipLaunchKernelGGL((ln_bwd_lean_kernel<2, false>), g, dim3(256), 0, s, DY, X, G, mean, rstd, A, (bf16_t*)dx, dgamma_part, dbeta_part, rows, C,
                       mdy, mx, mdx, madd, dr, (act_gelu & IFSEG_LN_PARAMS_F32) ? 1 : 0, (bf16_t*)nullptr, RowMap{}, DropArgs{});
  else if (C <= 1024) launch_ln_bwd<2, 1>(act_gelu, g, s, DY, X, G, mean, rstd, A, (bf16_t*)dx, dgamma_part, dbeta_part, rows, C, mdy, mx, mdx, madd, dr);
  else launch_ln_bwd<2, 4>(act_gelu, g, s, DY, X, G, mean, rstd, A, (bf16_t*)dx, dgamma_part, dbeta_part, rows, C, mdy, mx, mdx, madd, dr);
  ifseg_prof_end(IFSEG_K_LN_BWD, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_ln_bwd_drop(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                 const void* dx_add, void* dx, float* dgamma_part, float* dbeta_part, void* dx2, int nblocks,
                                 int rows, int C, int flags, int rpb, long long dy_bs, int lddy, long long x_bs, int ldx,
                                 long long dx_bs, int lddx, long long add_bs, int ldadd, long long dx2_bs, int lddx2,
                                 const ifseg_drop_args* drop2, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if ((C & 7) || C > 1024 || nblocks <= 0) return IFSEG_ERR_BAD_SHAPE;
  if (!dx2) return IFSEG_ERR_BAD_ARG;
  if (((lddy | ldx | lddx | lddx2) & 7) || (dx_add && (ldadd & 7))) return IFSEG_ERR_BAD_SHAPE;
  DropArgs dr{};
  if (drop2) {
    if (drop2->p < 0.f || drop2->p >= 1.f || drop2->rows_per_batch <= 0) return IFSEG_ERR_BAD_ARG;
    dr = DropArgs{1, drop2->p, drop2->seed, drop2->drop_path_scale, drop2->rows_per_batch, drop2->seed_add};
  }
  RowMap mdy = row_map(rpb, dy_bs, lddy), mx = row_map(rpb, x_bs, ldx), mdx = row_map(rpb, dx_bs, lddx), madd = row_map(rpb, add_bs, ldadd), mdx2 = row_map(rpb, dx2_bs, lddx2);
  hipStream_t s = (hipStream_t)stream;
  ifseg_prof_begin(IFSEG_K_LN_BWD, s, 0, (double)rows * C * (dx_add ? 10.0 : 8.0));
  hipLaunchKernelGGL((ln_bwd_lean_kernel<2, true>), dim3(nblocks), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dx_add, (bf16_t*)dx, dgamma_part, dbeta_part, rows, C,
                     mdy, mx, mdx, madd, DropArgs{}, (flags & IFSEG_LN_PARAMS_F32) ? 1 : 0, (bf16_t*)dx2, mdx2, dr);
  ifseg_prof_end(IFSEG_K_LN_BWD, s);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

namespace {
// master[i] <- p16[i] wherever the bf16 arena no longer is the rounding of the fp32 master (an external optimizer wrote
// the bf16 parameters); entries that still agree keep their full-precision value
__global__ void sync_master_kernel(float* master, const bf16_t* p16, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bf16_t h = p16[i];
    if (f2bf(master[i]) != h) master[i] = bf2f(h);
  }
}
}  // namespace

extern "C" int ifseg_sync_master(float* master, const void* p16, long long n, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sync_master_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, master, (const bf16_t*)p16, n);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_reduce_parts(const float* in, void* out, int outer, int parts, long long n, int accumulate,
                                  int out_bf16, float scale, void* stream) {
  (void)hipGetLastError();
  const long long total = (long long)outer * n;
  if (total <= 0) return 0;
  if (parts >= 2048 && total <= 4096) {
    if (out_bf16) hipLaunchKernelGGL(reduce_parts_tall_kernel<true>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
    else hipLaunchKernelGGL(reduce_parts_tall_kernel<false>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
    IFSEG_CHECK_LAUNCH();
    return 0;
  }
  if (parts >= 32) {
    dim3 g2((unsigned)(outer * ((n + 31) / 32)));
    if (out_bf16) hipLaunchKernelGGL(reduce_parts2d_kernel<true>, g2, dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
    else hipLaunchKernelGGL(reduce_parts2d_kernel<false>, g2, dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
    IFSEG_CHECK_LAUNCH();
    return 0;
  }
  dim3 g((unsigned)((total + 255) / 256));
  if (out_bf16) hipLaunchKernelGGL(reduce_parts_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
  else hipLaunchKernelGGL(reduce_parts_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, in, out, outer, parts, n, accumulate, scale);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_reduce_parts_multi(int ntask, const ifseg_reduce_task* tasks, void* stream) {
  (void)hipGetLastError();
  if (ntask <= 0) return 0;
  if (ntask > 16 || !tasks) return IFSEG_ERR_BAD_ARG;
  ReduceTasks ts{};
  ts.n = ntask;
  long long total = 0;
  for (int i = 0; i < ntask; ++i) {
    if (!tasks[i].in || !tasks[i].out || tasks[i].outer <= 0 || tasks[i].parts <= 0 || tasks[i].n <= 0) return IFSEG_ERR_BAD_ARG;
    ts.t[i] = tasks[i];
    ts.start[i] = (int)total;
    total += (long long)tasks[i].outer * ((tasks[i].n + 31) / 32);
    if (total >= (1ll << 31)) return IFSEG_ERR_BAD_SHAPE;
  }
  ts.start[ntask] = (int)total;
  hipLaunchKernelGGL(reduce_parts_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, ts);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_colsum_bf16(const void* x, float* part, int nblk_rows, int M, int N, int rpb, long long x_bs,
                                 int ldx, void* stream) {
  (void)hipGetLastError();
  if (M <= 0) return 0;
  if (N & 7) return IFSEG_ERR_BAD_SHAPE;
  RowMap mx = row_map(rpb, x_bs, ldx);
  const int rows_per_blk = (M + nblk_rows - 1) / nblk_rows;
  dim3 g((N / 8 + 255) / 256, nblk_rows);
  hipLaunchKernelGGL(colsum_kernel, g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, part, M, N, mx, rows_per_blk);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

namespace {
// gw[n][c] -= db[n] * xmean[c]
__global__ __launch_bounds__(256) void kproj_common_mode_kernel(bf16_t* gw, const bf16_t* db, const float* xmean, int N, int C) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;       // C % 8 == 0: eight columns of one row
  if (i >= (long long)N * C) return;
  const int n = (int)(i / C), c = (int)(i - (long long)n * C);
  const float d = bf2f(db[n]);
  float g[8];
  unpack8(*reinterpret_cast<const uint4*>(gw + i), g);
  const float4 m0 = *reinterpret_cast<const float4*>(xmean + c), m1 = *reinterpret_cast<const float4*>(xmean + c + 4);
  g[0] -= d * m0.x; g[1] -= d * m0.y; g[2] -= d * m0.z; g[3] -= d * m0.w;
  g[4] -= d * m1.x; g[5] -= d * m1.y; g[6] -= d * m1.z; g[7] -= d * m1.w;
  *reinterpret_cast<uint4*>(gw + i) = pack8(g);
}
}  // namespace

// The weight gradient of a KEY projection with the token-common component of its input removed:
//     gw <- gw - db (x) mean_rows(x),      gw [N, C] = dK^T x (bf16), db [N] = sum over all rows of dK (the bias gradient).
// Exact identity: softmax is invariant to adding one vector to every key of a (batch, head), so sum_j dK_j = 0 and
// dK^T x = dK^T (x - 1 c^T) for every c -- the reference's k_proj.bias gradient is float noise for the same reason
// (unify_multihead_attention.py:327-346 under autograd).  In bf16 arithmetic sum_j dK_j is NOT zero (delta = rowsum(dO * O)
// uses the bf16-rounded O, so the rows of dS do not sum to zero exactly), and that spurious sum multiplies the mean of x, which
// is 4-8 x larger than x's token-dependent part after a LayerNorm with a bias: measured on SegOFA-Base (tools/kproj_err.py,
// encoder layer 5) k_proj.weight rel-L2 against the fp32 reference 0.119 as computed, 0.017 with this rank-1 term removed.
extern "C" int ifseg_kproj_common_mode(void* gw, const void* db, const float* xmean, int N, int C, void* stream) {
  (void)hipGetLastError();
  if (!gw || !db || !xmean || N <= 0 || C <= 0 || (C & 7) || ((size_t)gw & 15) || ((size_t)xmean & 15)) return IFSEG_ERR_BAD_ARG;
  const long long threads = (long long)N * C / 8;
  hipLaunchKernelGGL(kproj_common_mode_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)gw, (const bf16_t*)db, xmean, N, C);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_embed_rows(const void* table, const long long* ids, const void* add, void* out, int n, int C,
                                int rpb, long long o_bs, int ldo, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  if (C & 7) return IFSEG_ERR_BAD_SHAPE;
  RowMap mo = row_map(rpb, o_bs, ldo);
  const long long total = (long long)n * (C / 8);
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)table, ids, (const bf16_t*)add, (bf16_t*)out, n, C, mo);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// Image-free patch embeddings: out[b, p, :] = mean_{t in bag(b,p)} table[ids[b, t], :] + add[:]
// (nn.EmbeddingBag(mode='mean') over the token table, encoder_module.py:147-148,529-538, followed by the
// type embedding of the image segment :589-591).  Bags are described the way the collater ships them:
// ids [B, maxlen] (padded at the end of each row), ends [B, P] = per-sample cumulative bag ends, so bag
// (b, p) = ids[b, ends[b,p-1] : ends[b,p]] -- no pad stripping, no host sync.  Empty bag -> add only.
namespace {
__global__ void embed_bag_mean_kernel(const bf16_t* table, const long long* ids, const long long* ends, const bf16_t* add,
                                      bf16_t* out, int B, int P, int C, int maxlen, RowMap mo) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C >> 3;
  if (gid >= (long long)B * P * nch) return;
  const int c = (int)(gid % nch) * 8;
  const int bag = (int)(gid / nch), b = bag / P, p = bag - b * P;
  const long long lo = p ? ends[(long long)b * P + p - 1] : 0, hi = ends[(long long)b * P + p];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long t = lo; t < hi && t < maxlen; ++t) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(table + ids[(long long)b * maxlen + t] * C + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  const float inv = hi > lo ? 1.f / (float)(hi - lo) : 0.f;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (add) unpack8(*reinterpret_cast<const uint4*>(add + c), a);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = acc[e] * inv + a[e];
  *reinterpret_cast<uint4*>(out + mo.off(bag) + c) = pack8(acc);
}
}  // namespace

extern "C" int ifseg_embed_bag_mean(const void* table, const long long* ids, const long long* ends, const void* add,
                                    void* out, int B, int P, int C, int maxlen, int rpb, long long o_bs, int ldo,
                                    void* stream) {
  (void)hipGetLastError();
  if (B <= 0 || P <= 0) return 0;
  if ((C & 7) || maxlen <= 0) return IFSEG_ERR_BAD_SHAPE;
  RowMap mo = row_map(rpb, o_bs, ldo);
  const long long total = (long long)B * P * (C / 8);
  hipLaunchKernelGGL(embed_bag_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)table, ids, ends, (const bf16_t*)add, (bf16_t*)out, B, P, C, maxlen, mo);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// Gradient of an embedding look-up (nn.Embedding / nn.EmbeddingBag(mode='mean') backward: the token table when
// --freeze-encoder-embedding / --freeze-decoder-embedding are false, unify_transformer.py:362-371): entries sorted by token id
// (stable: the caller's torch.sort), entry j contributes weight[j] * src[src_row[j], :] to row sorted_ids[j] of the table
// gradient.  One workgroup per entry; the FIRST entry of a run of equal ids sums the whole run in order and adds it to the
// table row -- fixed summation order, no atomics, no data-dependent launch shape.  ids outside [0, V) are skipped.
namespace {
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(const bf16_t* src, const long long* src_row, const float* weight,
                                                               const long long* sorted_ids, bf16_t* table_grad, long long m,
                                                               int C, long long V, long long skip_id) {
  const long long j = blockIdx.x;
  const long long id = sorted_ids[j];
  if (id < 0 || id >= V || id == skip_id) return;
  if (j > 0 && sorted_ids[j - 1] == id) return;
  for (int c = threadIdx.x * 8; c < C; c += blockDim.x * 8) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long jj = j; jj < m && sorted_ids[jj] == id; ++jj) {
      const float w = weight ? weight[jj] : 1.f;
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(src + src_row[jj] * C + c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, f[e], acc[e]);
    }
    float g[8];
    bf16_t* gp = table_grad + id * C + c;
    unpack8(*reinterpret_cast<const uint4*>(gp), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] += acc[e];
    *reinterpret_cast<uint4*>(gp) = pack8(g);
  }
}
}  // namespace

extern "C" int ifseg_rows_segment_sum(const void* src, const long long* src_row, const float* weight, const long long* sorted_ids,
                                      void* table_grad, long long m, int C, long long V, long long skip_id, void* stream) {
  (void)hipGetLastError();
  if (m <= 0) return 0;
  if ((C & 7) || !src || !src_row || !sorted_ids || !table_grad || V <= 0 || m >= (1ll << 31)) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(rows_segment_sum_kernel, dim3((unsigned)m), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_row,
                     weight, sorted_ids, (bf16_t*)table_grad, m, C, V, skip_id);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_cast_f32_bf16(const float* in, void* out, long long n, float scale, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, n, scale);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_add_bf16(const void* a, const void* b, void* out, long long n, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_nchw_to_nhwc_bf16(const void* in, int in_is_f32, void* out, int B, int C, int H, int W, int Cpad,
                                       void* stream) {
  (void)hipGetLastError();
  const long long total = (long long)B * H * W * Cpad;
  if (total <= 0) return 0;
  dim3 g((unsigned)((total + 255) / 256));
  if (in_is_f32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)in, (bf16_t*)out, B, C, H, W, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, B, C, H, W, Cpad);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// ---- input validation without torch glue -----------------------------------------------------------
// One launch per check, the verdict written straight into PINNED host memory (the engine reads it once the event behind the
// launch has completed: HipEngine.deferred_check).  The torch composition it replaces -- compare, reduce, cast, copy -- was four
// tiny kernels per check, three checks per step, all on the main queue between two steps.
//   mode 0: any x[i] == value                      (int64: a <pad> source token where none is allowed, encoder_module.py:730-752)
//   mode 1: rows of `row_len`: any x == value followed by x != value, or x[row start] == value  (padding that is not a suffix)
//   mode 2: any x[i] == 0, x as bytes                (bool: a masked-out patch image, encoder_module.py:388-404 `patch_masks`)
//   mode 3: *flag != 0, then *flag = 0               (int32 device flag raised by a kernel: the loss kernel's bad-label flag)
namespace {
__global__ __launch_bounds__(256) void check_inputs_kernel(const void* x, long long n, int mode, long long value, long long row_len,
                                                           unsigned char* verdict) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int bad = 0;
  if (mode == 3) {
    if (threadIdx.x == 0) {
      int* f = reinterpret_cast<int*>(const_cast<void*>(x));
      bad = *f != 0;
      *f = 0;
    }
  } else {
    for (long long i = threadIdx.x; i < n; i += 256) {
      if (mode == 2) {
        bad |= reinterpret_cast<const unsigned char*>(x)[i] == 0;
      } else {
        const long long* t = reinterpret_cast<const long long*>(x);
        if (mode == 0) bad |= t[i] == value;
        else {
          const long long c = i % row_len;
          bad |= (c == 0 && t[i] == value) || (c + 1 < row_len && t[i] == value && t[i + 1] != value);
        }
      }
    }
  }
  if (bad) any = 1;          // (benign race: every writer stores the same value)
  __syncthreads();
  if (threadIdx.x == 0) *verdict = any ? 1 : 0;
}
}  // namespace

extern "C" int ifseg_check_inputs(const void* x, long long n, int mode, long long value, long long row_len,
                                  unsigned char* verdict_host_pinned, void* stream) {
  (void)hipGetLastError();
  if (!x || !verdict_host_pinned || mode < 0 || mode > 3 || n < 0 || (mode == 1 && row_len <= 0)) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(check_inputs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, mode, value, row_len, verdict_host_pinned);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// ---- rel-pos table <-> per-head delta tables ------------------------------------
namespace {
// out[h][i] = table[idx[i]][h] (idx < 0 -> 0)
__global__ void rel_gather_kernel(const bf16_t* table, const int* idx, float* out, int n, int H) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * H) return;
  const int h = gid / n, i = gid % n;
  const int r = idx[i];
  out[gid] = r >= 0 ? bf2f(table[(long long)r * H + h]) : 0.f;
}
// the same gather for up to 16 tables of one shape at once (the per-layer rel-pos tables): out[l][h][i]
struct TablePtrs { const bf16_t* t[16]; };
__global__ void rel_gather_multi_kernel(TablePtrs tabs, const int* idx, float* out, int n, int H, int L) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)L * n * H) return;
  const int i = (int)(gid % n), h = (int)((gid / n) % H), l = (int)(gid / ((long long)n * H));
  const int r = idx[i];
  out[gid] = r >= 0 ? bf2f(tabs.t[l][(long long)r * H + h]) : 0.f;
}
// acc[idx[i]][h] += d[h][i]   (fp32 accumulation buffer, tiny; duplicates allowed)
__global__ void rel_scatter_kernel(const float* d, const int* idx, float* acc, int n, int H) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * H) return;
  const int h = gid / n, i = gid % n;
  const int r = idx[i];
  if (r >= 0) atomicAdd(&acc[(long long)r * H + h], d[gid]);
}
}  // namespace

extern "C" int ifseg_rel_gather(const void* table, const int* idx, float* out, int n, int H, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rel_gather_kernel, dim3((n * H + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)table, idx, out, n, H);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
extern "C" int ifseg_rel_gather_multi(const void* const* tables, int L, const int* idx, float* out, int n, int H,
                                      void* stream) {
  (void)hipGetLastError();
  if (n <= 0 || L <= 0) return 0;
  if (L > 16) return IFSEG_ERR_BAD_ARG;
  TablePtrs tp{};
  for (int l = 0; l < L; ++l) tp.t[l] = (const bf16_t*)tables[l];
  const long long total = (long long)L * n * H;
  hipLaunchKernelGGL(rel_gather_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tp, idx,
                     out, n, H, L);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
extern "C" int ifseg_rel_scatter_add(const float* d, const int* idx, float* acc, int n, int H, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rel_scatter_kernel, dim3((n * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, idx, acc, n, H);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// stand-alone dropout + DropPath: out = [resid +] dpscale[b] * keep(i) * x / (1 - p)  (mask: drop8 above;
// the backward is the same call with x = dy, resid = NULL)
namespace {
__global__ void dropout_kernel(const bf16_t* x, const bf16_t* resid, bf16_t* out, long long nchunks, int C, float p,
                               unsigned long long seed, const float* dpscale, int rows_per_batch, RowMap mx, RowMap mr,
                               RowMap mo, const unsigned long long* seed_add) {
  const long long c8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c8 >= nchunks) return;
  const int nch = C >> 3;
  const int row = (int)(c8 / nch), col = (int)(c8 % nch) * 8;
  float f[8];
  unpack8(*reinterpret_cast<const uint4*>(x + mx.off(row) + col), f);
  drop8(f, DropArgs{1, p, seed, dpscale, rows_per_batch, seed_add}, c8, row);
  if (resid) {
    float r[8];
    unpack8(*reinterpret_cast<const uint4*>(resid + mr.off(row) + col), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += r[e];
  }
  *reinterpret_cast<uint4*>(out + mo.off(row) + col) = pack8(f);
}
}  // namespace

extern "C" int ifseg_dropout(const void* x, const void* resid, void* out, long long rows, int C, float p,
                             unsigned long long seed, const float* drop_path_scale, int rows_per_batch, int rpb,
                             long long x_bs, int ldx, long long r_bs, int ldr, long long o_bs, int ldo,
                             const unsigned long long* seed_add, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if ((C & 7) || p < 0.f || p >= 1.f || rows_per_batch <= 0) return IFSEG_ERR_BAD_ARG;
  const long long nchunks = rows * (C / 8);
  RowMap mx = row_map(rpb, x_bs, ldx), mr = row_map(rpb, r_bs, ldr), mo = row_map(rpb, o_bs, ldo);
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)resid, (bf16_t*)out, nchunks, C, p, seed, drop_path_scale, rows_per_batch,
                     mx, mr, mo, seed_add);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// out = keep ? x : fill (NO rescaling) with ifseg_dropout's mask for (seed, element): the activation dropout between GELU and the
// FFN LayerNorm (unify_transformer_layer.py:280,556) is applied to the PRE-activation u, in place, with fill = -30:
// gelu(-30) = -0 and gelu'(-30) = 0 exactly in fp32 (exp(-450) underflows, the erf polynomial returns -1), so every kernel
// that recomputes gelu(u) -- LayerNorm forward / backward, the fused GEMM epilogue, the ffn_ln gradient kernels -- sees
// a = keep * gelu(u) and da/du = keep * gelu'(u) without knowing about the mask.  The missing factor 1 / (1 - p) cancels in the
// LayerNorm that follows: LN(a / (1 - p); eps) == LN(a; eps (1 - p)^2), forward and backward (the caller passes that eps).
// (A pre-activation that is LEGITIMATELY <= -30 is indistinguishable from a dropped one -- and needs no distinction: both give
// gelu = gelu' = 0 exactly, which is all any consumer reads from u.)
namespace {
__global__ void dropout_fill_kernel(const bf16_t* x, bf16_t* out, long long nchunks, float p, unsigned long long seed,
                                    const unsigned long long* seed_add, float fill) {
  const long long c8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c8 >= nchunks) return;
  float f[8], k[8];
  unpack8(*reinterpret_cast<const uint4*>(x + c8 * 8), f);
#pragma unroll
  for (int e = 0; e < 8; ++e) k[e] = 1.f;
  drop8(k, DropArgs{1, p, seed, nullptr, 1, seed_add}, c8, 0);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = k[e] != 0.f ? f[e] : fill;
  *reinterpret_cast<uint4*>(out + c8 * 8) = pack8(f);
}
}  // namespace

extern "C" int ifseg_dropout_fill(const void* x, void* out, long long n, float p, unsigned long long seed,
                                  const unsigned long long* seed_add, float fill, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  if ((n & 7) || p < 0.f || p >= 1.f || !x || !out) return IFSEG_ERR_BAD_ARG;
  const long long nchunks = n / 8;
  hipLaunchKernelGGL(dropout_fill_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, nchunks, p, seed, seed_add, fill);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

// DropPath keep masks (unify_transformer_layer.py:19-35): out[i][b] = Bernoulli(keep[i]) / keep[i] for residual branch i
// and sample b, from the same counter-based generator as the dropout masks (replay-safe: no host RNG state)
namespace {
__global__ void droppath_scale_kernel(float* out, const float* keep, int n, int B, unsigned long long seed,
                                      const unsigned long long* seed_add) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * B) return;
  const unsigned long long sd = seed + (seed_add ? *seed_add : 0ull);
  const float u = (float)(splitmix64(sd + 0x5851F42D4C957F2Dull * (unsigned long long)(gid + 1)) >> 40) * (1.f / 16777216.f);
  const float k = keep[gid / B];
  out[gid] = (u < k) ? 1.f / k : 0.f;
}
}  // namespace

extern "C" int ifseg_droppath_scale(float* out, const float* keep, int n, int B, unsigned long long seed,
                                    const unsigned long long* seed_add, void* stream) {
  (void)hipGetLastError();
  if (n * B <= 0) return 0;
  hipLaunchKernelGGL(droppath_scale_kernel, dim3((n * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, keep, n, B, seed, seed_add);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
