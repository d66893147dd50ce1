// ffn_layernorm backward folded into the fc2 dX GEMM: the small reduction kernels around the GEMM epilogue (csrc/gemm.hip, EPI_GLN).
//
// BUILD NOTE -- this file is compiled with -fno-slp-vectorize (ifseg_amd/build.py, PER_FILE_FLAGS).  clang's SLP vectoriser
// turns the two-accumulator loop of ffn_ln_coef_kernel (sum w*gamma next to sum w*beta) into packed-fp32 code that starts with
//     v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]        (low lane = A.lo * B.HI, high lane = A.hi * B.lo)
// and with that instruction the kernel returned a wrong LOW-half sum (sum w*gamma, never sum w*beta) in a few of its 9216 waves
// per launch whenever an LDS-DMA + MFMA GEMM (ours or hipBLASLt's) ran next to it on another stream -- bit-exact alone, and
// bit-exact under the same load in every form without that instruction (tools/probe/README.md, DESIGN.md "Round 3" (6)).
// tools/check_isa.py / test_library_has_no_cross_half_packed_fp32_instruction keep the form out of the built library;
// tests/test_kernels_gpu.py keeps these kernels under a concurrent GEMM and demands bit equality.
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// The FFN's ffn_layernorm(gelu(fc1)) on the way back WITHOUT a 3072-wide LayerNorm-backward pass
// (unify_transformer_layer.py:279-283 under autograd).  With z = gamma * xh + beta the input of fc2 (t = z W2^T + b2 its
// output, dY the gradient of t) the LayerNorm backward needs two row means over the N = 3072 columns of dz = dY W2:
//     c1 = mean_k(gamma_k dz_k)        = (1/N) sum_j dY_j a_j ,            a_j  = sum_k gamma_k W2[j,k]
//     c2 = mean_k(gamma_k xh_k dz_k)   = (1/N) sum_j dY_j (t_j - wb_j) ,   wb_j = b2_j + sum_k beta_k W2[j,k]
// (both are linear in dz, and sum_k gamma_k xh_k W2[j,k] is the forward product minus its beta / bias part) -- row dots over
// the 768 columns of dY and of the SAVED fc2 output.  With c1, c2 known per row, du is element-wise in dz and rides in the
// epilogue of the dX GEMM (csrc/gemm.hip, EPI_GLN): dz is never written, the wide LayerNorm-backward kernel (97 us in the
// step, x 12 layers) is gone.  The parameter gradients follow from the weight gradient of fc2, which is computed anyway:
//     dbeta_k  = sum_j db2_j W2[j,k]
//     dgamma_k = (sum_j W2[j,k] dW2[j,k] - beta_k dbeta_k) / gamma_k        (dW2[j,k] = sum_r dY[r,j] (gamma_k xh[r,k] + beta_k))
namespace {

// coef[0][j] = a_j, coef[1][j] = wb_j : one wave per row j of W2 [J, N]; blockIdx.y = layer (up to 32 layers per launch)
struct FfnCoefPtrs { const bf16_t* w2[32]; const float* gamma[32]; const float* beta[32]; const bf16_t* b2[32]; float* coef[32]; };
__global__ __launch_bounds__(256) void ffn_ln_coef_kernel(FfnCoefPtrs pt, int ldw, int J, int N) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= J) return;
  const bf16_t* __restrict__ w2 = pt.w2[blockIdx.y];
  const float* __restrict__ gamma = pt.gamma[blockIdx.y];
  const float* __restrict__ beta = pt.beta[blockIdx.y];
  const bf16_t* __restrict__ b2 = pt.b2[blockIdx.y];
  float* __restrict__ coef = pt.coef[blockIdx.y];
  const bf16_t* row = w2 + (long long)j * ldw;
  float sa = 0.f, sb = 0.f;
  for (int k = lane * 8; k < N; k += 64 * 8) {
    float w[8];
    unpack8(*reinterpret_cast<const uint4*>(row + k), w);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + k), g1 = *reinterpret_cast<const float4*>(gamma + k + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + k), b1 = *reinterpret_cast<const float4*>(beta + k + 4);
    sa += w[0] * g0.x + w[1] * g0.y + w[2] * g0.z + w[3] * g0.w + w[4] * g1.x + w[5] * g1.y + w[6] * g1.z + w[7] * g1.w;
    sb += w[0] * b0.x + w[1] * b0.y + w[2] * b0.z + w[3] * b0.w + w[4] * b1.x + w[5] * b1.y + w[6] * b1.z + w[7] * b1.w;
  }
  sa = warp_sum(sa); sb = warp_sum(sb);
  if (lane == 0) { coef[j] = sa; coef[J + j] = sb + (b2 ? bf2f(b2[j]) : 0.f); }
}

// c[r][0] = (1/N) sum_j dY[r][j] a_j ; c[r][1] = (1/N) sum_j dY[r][j] (t[r][j] - wb_j) : one wave per row
__global__ __launch_bounds__(256) void ffn_ln_rowstats_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ t,
                                                              int ldt, const float* __restrict__ coef, float* __restrict__ c,
                                                              int rows, int J, float inv_n) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int j = lane * 8; j < J; j += 64 * 8) {
    float d[8], tv[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + (long long)r * lddy + j), d);
    unpack8(*reinterpret_cast<const uint4*>(t + (long long)r * ldt + j), tv);
    const float4 a0 = *reinterpret_cast<const float4*>(coef + j), a1 = *reinterpret_cast<const float4*>(coef + j + 4);
    const float4 w0 = *reinterpret_cast<const float4*>(coef + J + j), w1 = *reinterpret_cast<const float4*>(coef + J + j + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += d[e] * av[e]; s2 += d[e] * (tv[e] - wv[e]); }
  }
  s1 = warp_sum(s1); s2 = warp_sum(s2);
  if (lane == 0) { c[2 * r] = s1 * inv_n; c[2 * r + 1] = s2 * inv_n; }
}

// stage 1: partial column sums over a slab of rows.  Block = 128 columns x 16 row lanes (a thread: 8 columns of every 16th
// row of its slab), grid (N / 128, PG_SLABS): part[slab][0][k] = sum_j W2[j,k] dW2[j,k], part[slab][1][k] = sum_j W2[j,k] db2_j
constexpr int PG_SLABS = 8;
__global__ __launch_bounds__(256) void ffn_ln_pg_partial_kernel(const bf16_t* __restrict__ w2, const bf16_t* __restrict__ dw2,
                                                                const bf16_t* __restrict__ db2, float* __restrict__ part, int J, int N) {
  __shared__ float red[2][16][128 + 4];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int k0 = blockIdx.x * 128 + cx * 8;
  const int rows_per = (J + PG_SLABS - 1) / PG_SLABS, j0 = blockIdx.y * rows_per, j1 = min(J, j0 + rows_per);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (k0 < N) {
    for (int j = j0 + ry; j < j1; j += 16) {
      float w[8], d[8];
      unpack8(*reinterpret_cast<const uint4*>(w2 + (long long)j * N + k0), w);
      unpack8(*reinterpret_cast<const uint4*>(dw2 + (long long)j * N + k0), d);
      const float dbj = bf2f(db2[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] += w[e] * d[e]; b[e] += w[e] * dbj; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][ry][cx * 8 + e] = a[e]; red[1][ry][cx * 8 + e] = b[e]; }
  __syncthreads();
  const int which = threadIdx.x >> 7, col = threadIdx.x & 127, k = blockIdx.x * 128 + col;
  if (k < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[which][r][col];          // fixed order: bit-reproducible
    part[((long long)blockIdx.y * 2 + which) * N + k] = t;
  }
}
// stage 2: dbeta_k = sum over slabs of part[.][1][k]; dgamma_k = (sum of part[.][0][k] - beta_k dbeta_k) / gamma_k
__global__ __launch_bounds__(256) void ffn_ln_pg_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, bf16_t* __restrict__ dgamma,
                                                              bf16_t* __restrict__ dbeta, int N) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  float swd = 0.f, sdb = 0.f;
#pragma unroll
  for (int sl = 0; sl < PG_SLABS; ++sl) { swd += part[((long long)sl * 2) * N + k]; sdb += part[((long long)sl * 2 + 1) * N + k]; }
  dbeta[k] = f2bf(sdb);
  // (a gain too small for this division -- see ffn_ln_gain_is_small -- is overwritten by ffn_ln_dgamma_exact_kernel; without
  // the operands of that rescue the result is kept finite)
  const float g = gamma[k];
  dgamma[k] = f2bf(g != 0.f ? (swd - beta[k] * sdb) / g : 0.f);
}

// Columns whose gain is too small for the division above (gamma_k = 0 gives 0 / 0; for |gamma_k| << |beta_k| the 2^-9
// rounding of dW2 is amplified by 1 / gamma_k): dgamma_k from its definition,
//     dgamma_k = sum_m dz[m][k] xhat[m][k],   dz[m][k] = sum_j dY[m][j] W2[j][k],  xhat = (gelu(u[m][k]) - mean_m) rstd_m,
// one column at a time by the block that owns it (13 MB of dY per column: a rescue path -- with no flagged column the
// launch is a dozen blocks that read their gammas and leave).  Same criterion on every run: deterministic.
__device__ __forceinline__ bool ffn_ln_gain_is_small(float g, float b) { return fabsf(g) < 0.05f * fmaxf(1.f, fabsf(b)); }
__global__ __launch_bounds__(256) void ffn_ln_dgamma_exact_kernel(const bf16_t* __restrict__ w2, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const bf16_t* __restrict__ dy,
                                                                  int lddy, const bf16_t* __restrict__ u, int ldu,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  bf16_t* __restrict__ dgamma, int M, int J, int N) {
  __shared__ float sw[1024];
  __shared__ float red[256];
  __shared__ int flagged[256];
  __shared__ int nflag;
  const int tid = threadIdx.x;
  // every thread tests one column; the (normally empty) list of flagged columns is walked in column order
  if (tid == 0) nflag = 0;
  __syncthreads();
  {
    const int k = blockIdx.x * 256 + tid;
    const bool f = k < N && ffn_ln_gain_is_small(gamma[k], beta[k]);
    flagged[tid] = f ? 1 : 0;
    if (f) atomicAdd(&nflag, 1);
  }
  __syncthreads();
  if (nflag == 0) return;
  for (int kk = 0; kk < 256; ++kk) {
    if (!flagged[kk]) continue;          // (block-uniform)
    const int k = blockIdx.x * 256 + kk;
    __syncthreads();
    for (int j = tid; j < J; j += 256) sw[j] = bf2f(w2[(long long)j * N + k]);
    __syncthreads();
    float acc = 0.f;
    for (int m = tid; m < M; m += 256) {
      float dz = 0.f;
      const bf16_t* dp = dy + (long long)m * lddy;
      for (int j = 0; j < J; j += 8) {
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dp + j), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) dz += d[e] * sw[j + e];
      }
      const float uv = bf2f(u[(long long)m * ldu + k]);
      const float g = 0.5f * uv * (1.f + erf_as(uv, __expf(-0.5f * uv * uv)));
      acc += dz * (g - mean[m]) * rstd[m];
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) dgamma[k] = f2bf(red[0]);
  }
}

}  // namespace

extern "C" int ifseg_ffn_ln_coef(const void* const* w2, int ldw, const float* const* gamma, const float* const* beta,
                                 const void* const* b2, float* const* coef, int L, int J, int N, void* stream) {
  (void)hipGetLastError();
  if (!w2 || !gamma || !beta || !coef || L <= 0 || L > 32 || J <= 0 || N <= 0 || (N & 7) || (ldw & 7)) return IFSEG_ERR_BAD_ARG;
  FfnCoefPtrs pt{};
  for (int l = 0; l < L; ++l) {
    if (!w2[l] || !gamma[l] || !beta[l] || !coef[l] || (((size_t)gamma[l] | (size_t)beta[l]) & 15)) return IFSEG_ERR_BAD_ARG;
    pt.w2[l] = (const bf16_t*)w2[l]; pt.gamma[l] = gamma[l]; pt.beta[l] = beta[l];
    pt.b2[l] = b2 ? (const bf16_t*)b2[l] : nullptr; pt.coef[l] = coef[l];
  }
  hipLaunchKernelGGL(ffn_ln_coef_kernel, dim3((J + 3) / 4, L), dim3(256), 0, (hipStream_t)stream, pt, ldw, J, N);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_ffn_ln_rowstats(const void* dy, int lddy, const void* t, int ldt, const float* coef, float* c, int rows,
                                     int J, int N, void* stream) {
  (void)hipGetLastError();
  if (!dy || !t || !coef || !c || rows <= 0 || J <= 0 || N <= 0 || (J & 7) || (lddy & 7) || (ldt & 7) || ((size_t)coef & 15))
    return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(ffn_ln_rowstats_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy,
                     (const bf16_t*)t, ldt, coef, c, rows, J, 1.f / (float)N);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_ffn_ln_param_grads(const void* w2, const void* dw2, const void* db2, const float* gamma, const float* beta,
                                        void* dgamma, void* dbeta, float* workspace /* >= 16 N floats */, int J, int N,
                                        const void* dy, int lddy, const void* u, int ldu, const float* mean, const float* rstd,
                                        int M, void* stream) {
  (void)hipGetLastError();
  if (!w2 || !dw2 || !db2 || !gamma || !beta || !dgamma || !dbeta || !workspace || J <= 0 || N <= 0 || (N & 7)) return IFSEG_ERR_BAD_ARG;
  if (dy && (!u || !mean || !rstd || M <= 0 || (J & 7) || J > 1024 || (lddy & 7))) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(ffn_ln_pg_partial_kernel, dim3((N + 127) / 128, PG_SLABS), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w2, (const bf16_t*)dw2, (const bf16_t*)db2, workspace, J, N);
  hipLaunchKernelGGL(ffn_ln_pg_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, gamma, beta,
                     (bf16_t*)dgamma, (bf16_t*)dbeta, N);
  if (dy)
    hipLaunchKernelGGL(ffn_ln_dgamma_exact_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w2, gamma,
                       beta, (const bf16_t*)dy, lddy, (const bf16_t*)u, ldu, mean, rstd, (bf16_t*)dgamma, M, J, N);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
