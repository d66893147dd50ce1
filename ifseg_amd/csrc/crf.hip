// Dense-CRF mean-field post-processing on gfx950 (BASELINE config 5; reference crf.py:19-37).
//
// The reference calls pydensecrf: unary -log(clip(p, 1e-5)), a spatial Gaussian pairwise term (sxy = 1, Potts 3), a
// bilateral one (sxy = 67, srgb = 3, Potts 4), symmetric normalisation n = 1/sqrt(K 1), 10 mean-field iterations
//     Q <- softmax(-U + sum_k w_k n_k .* (K_k (n_k .* Q))).
// pydensecrf evaluates K Q with the permutohedral lattice (an approximate filter).  MI355X formulation: the EXACT dense
// bilateral filter.  K (N x N, N = h w pixels, never stored) is generated tile by tile on the VALUs in fp32 --
//     k(i, j) = gx[xi - xj] * gy[yi - yj] * exp2(-|c_i - c_j|^2),   colours pre-scaled by sqrt(log2 e / (2 srgb^2)),
// a 64 x 64 tile of it is rounded to bf16 into LDS and multiplied with the class-major message operand Qn[c][j]
// (= n_j Q[c][j], bf16) on the MFMA units, fp32 accumulation over all j.  Cost per pass ~ N^2 kernel evaluations
// (6.9e10 for 512 x 512: ~10 ms); one pass with Qn = 1 yields the normalisation.  The sxy = 1 term is an 11 x 11 stencil.
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr int TI = 64, TJ = 64;
constexpr int GMAX = 2048;             // image sides up to 2048

// out[c][i] = sum_j k(i, j) * qn[c][j]   for c < Cp (Cp % 32 == 0), i < N
//   feat [N][4] fp32: (r', g', b', bits x | y << 16) pre-scaled colours + pixel coordinates;  gx / gy [GMAX] fp32: exp(-d^2 / (2 sxy^2)) for d = 0..
//   qn [Cp][ldq] bf16 (ldq >= N rounded up to 64, zero padded);  out [Cp][N] fp32
template <int CT>                       // CT = Cp / 32 class tiles per pixel tile (1..6)
__global__ __launch_bounds__(256, 2) void crf_bilateral_kernel(const float4* __restrict__ feat, const float* __restrict__ gx,
                                                               const float* __restrict__ gy, const bf16_t* __restrict__ qn,
                                                               int ldq, float* __restrict__ out, int N, int W) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[TI * TJ * 2];          // K tile  [i][j] bf16, 128-byte rows, swizzled
  __shared__ __attribute__((aligned(16))) unsigned char sQ[CT * 32 * TJ * 2];     // Qn tile [c][j] bf16
  __shared__ float sGx[GMAX], sGy[GMAX];
  __shared__ float4 sFj[TJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * TI;
  const int H = (N + W - 1) / W;
  for (int d = tid; d < W; d += 256) sGx[d] = gx[d];
  for (int d = tid; d < H; d += 256) sGy[d] = gy[d];
  // this thread evaluates row il = tid & 63 of the tile for the 16 columns jq*16 .. +15
  const int il = tid & 63, jq = tid >> 6;
  const int i = min(i0 + il, N - 1);
  const float4 fi = feat[i];
  const int xi = __float_as_int(fi.w) & 0xffff, yi = __float_as_int(fi.w) >> 16;      // pixel coordinates ride in .w
  // MFMA roles: wave -> pixel half (wave & 1), class tiles ct = (wave >> 1), (wave >> 1) + 2, ...
  constexpr int NT = (CT + 1) / 2;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  auto koff = [](int r, int c16) { return r * 128 + ((c16 ^ (r & 7)) << 4); };   // 8 chunks of 16 B per row
  const int ntile = (N + TJ - 1) / TJ;
  for (int jt = 0; jt < ntile; ++jt) {
    const int j0 = jt * TJ;
    __syncthreads();                                       // previous tile's MFMAs are done with sK / sQ / sFj
    if (tid < TJ) sFj[tid] = feat[min(j0 + tid, N - 1)];
    // Qn tile: CT*32 rows x 64 j bf16 = CT*32*8 chunks of 16 B
    for (int ch = tid; ch < CT * 32 * 8; ch += 256) {
      const int c = ch >> 3, c16 = ch & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(qn + (long long)c * ldq + j0 + c16 * 8);
      *reinterpret_cast<uint4*>(sQ + koff(c, c16)) = v;
    }
    __syncthreads();
    // ---- kernel values of this thread's 16 pairs
    {
      uint32_t pk[8];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        float kv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int jl = jq * 16 + e + u, j = j0 + jl;
          const float4 fj = sFj[jl];
          const int xj = __float_as_int(fj.w) & 0xffff, yj = __float_as_int(fj.w) >> 16;
          const float dr = fi.x - fj.x, dg = fi.y - fj.y, db = fi.z - fj.z;
          const float d2 = fmaf(dr, dr, fmaf(dg, dg, db * db));
          float k = __builtin_amdgcn_exp2f(-d2) * sGx[abs(xi - xj)] * sGy[abs(yi - yj)];
          kv[u] = (j < N) ? k : 0.f;
        }
        pk[e >> 1] = pack2bf(kv[0], kv[1]);
      }
      *reinterpret_cast<uint4*>(sK + koff(il, jq * 2)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      *reinterpret_cast<uint4*>(sK + koff(il, jq * 2 + 1)) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
    __syncthreads();
    // ---- out[c][i] += sum_j Qn[c][j] K[i][j]: D[n = i][m = c] with A = Qn rows (c), B = K rows (i)
    const int ih = wave & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ct = (wave >> 1) + 2 * t;
      if (ct < CT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 fa = lds_read_b128(sQ + koff(ct * 32 + (lane & 31), ks * 2 + (lane >> 5)));
          const bf16x8 fb = lds_read_b128(sK + koff(ih * 32 + (lane & 31), ks * 2 + (lane >> 5)));
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // D[row = c: (r&3) + 8*(r>>2) + 4*(lane>>5)][col = i: lane & 31]
  const int ih = wave & 1;
  const int io = i0 + ih * 32 + (lane & 31);
  if (io < N) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ct = (wave >> 1) + 2 * t;
      if (ct < CT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          out[(long long)c * N + io] = acc[t][r];
        }
      }
    }
  }
}

// spatial Gaussian (sxy): out[c][i] = sum_{|dx|,|dy| <= R} g[dx] g[dy] qn[c][i + (dx, dy)], qn fp32 [C][N]; C real classes
__global__ void crf_spatial_kernel(const float* __restrict__ qn, const float* __restrict__ g, int R, float* __restrict__ out,
                                   int C, int H, int W) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long N = (long long)H * W;
  if (gid >= N * C) return;
  const int c = (int)(gid / N), i = (int)(gid % N), x = i % W, y = i / W;
  const float* q = qn + (long long)c * N;
  float s = 0.f;
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    float row = 0.f;
    for (int dx = -R; dx <= R; ++dx) {
      const int xx = x + dx;
      if (xx >= 0 && xx < W) row += g[abs(dx)] * q[(long long)yy * W + xx];
    }
    s += g[abs(dy)] * row;
  }
  out[gid] = s;
}

// one mean-field update per pixel:  Q = softmax_c(-U + wp * npos * mpos + wb * nbi * mbi);
// writes Q [C][N] fp32, the spatial operand npos*Q (fp32 [C][N]) and the bilateral operand nbi*Q (bf16 [Cp][ldq]).
// mpos / mbi may be null (initialisation: Q = softmax(-U)).  U = -log(clip(prob, 1e-5, 1)) is formed here from `prob`.
__global__ void crf_update_kernel(const float* __restrict__ prob, const float* __restrict__ mpos, const float* __restrict__ mbi,
                                  const float* __restrict__ npos, const float* __restrict__ nbi, float wp, float wb,
                                  float* __restrict__ Q, float* __restrict__ qpos, bf16_t* __restrict__ qbi, int ldq, int C,
                                  int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float np_ = npos ? npos[i] : 1.f, nb_ = nbi ? nbi[i] : 1.f;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) {
    float v = __logf(fminf(fmaxf(prob[(long long)c * N + i], 1e-5f), 1.f));
    if (mpos) v += wp * np_ * mpos[(long long)c * N + i];
    if (mbi) v += wb * nb_ * mbi[(long long)c * N + i];
    Q[(long long)c * N + i] = v;
    m = fmaxf(m, v);
  }
  float s = 0.f;
  for (int c = 0; c < C; ++c) {
    const float e = __expf(Q[(long long)c * N + i] - m);
    Q[(long long)c * N + i] = e;
    s += e;
  }
  const float inv = 1.f / s;
  for (int c = 0; c < C; ++c) {
    const float q = Q[(long long)c * N + i] * inv;
    Q[(long long)c * N + i] = q;
    if (qpos) qpos[(long long)c * N + i] = q * np_;
    if (qbi) qbi[(long long)c * ldq + i] = f2bf(q * nb_);
  }
}

// n = 1 / sqrt(k1 + 1e-20) from the "K 1" pass (class row 0 of a filter output)
__global__ void crf_norm_kernel(const float* __restrict__ k1, float* __restrict__ n, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) n[i] = rsqrtf(k1[i] + 1e-20f);
}

}  // namespace

extern "C" int ifseg_crf_bilateral(const float* feat4, const float* gx, const float* gy, const void* qn, int ldq, float* out,
                                   int Cp, int H, int W, void* stream) {
  (void)hipGetLastError();
  const int N = H * W;
  if (N <= 0) return 0;
  if ((Cp & 31) || Cp < 32 || Cp > 192 || (ldq & 63) || ldq < ((N + 63) & ~63) || W > GMAX || H > GMAX) return IFSEG_ERR_BAD_SHAPE;
  dim3 g((N + TI - 1) / TI), b(256);
  hipStream_t s = (hipStream_t)stream;
  const float4* f = reinterpret_cast<const float4*>(feat4);
  const bf16_t* q = (const bf16_t*)qn;
  switch (Cp / 32) {
    case 1: hipLaunchKernelGGL(crf_bilateral_kernel<1>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
    case 2: hipLaunchKernelGGL(crf_bilateral_kernel<2>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
    case 3: hipLaunchKernelGGL(crf_bilateral_kernel<3>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
    case 4: hipLaunchKernelGGL(crf_bilateral_kernel<4>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
    case 5: hipLaunchKernelGGL(crf_bilateral_kernel<5>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
    default: hipLaunchKernelGGL(crf_bilateral_kernel<6>, g, b, 0, s, f, gx, gy, q, ldq, out, N, W); break;
  }
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_crf_spatial(const float* qn, const float* g, int R, float* out, int C, int H, int W, void* stream) {
  (void)hipGetLastError();
  const long long tot = (long long)C * H * W;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(crf_spatial_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qn, g, R, out, C, H, W);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_crf_update(const float* prob, const float* mpos, const float* mbi, const float* npos, const float* nbi,
                                float wpos, float wbi, float* Q, float* qpos, void* qbi, int ldq, int C, int N, void* stream) {
  (void)hipGetLastError();
  if (N <= 0) return 0;
  hipLaunchKernelGGL(crf_update_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, mpos, mbi, npos, nbi, wpos,
                     wbi, Q, qpos, (bf16_t*)qbi, ldq, C, N);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_crf_norm(const float* k1, float* n, int N, void* stream) {
  (void)hipGetLastError();
  if (N <= 0) return 0;
  hipLaunchKernelGGL(crf_norm_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, k1, n, N);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
