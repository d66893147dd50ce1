// Eval-time post-processing of the segmentation criterion on gfx950 (BASELINE config 5):
//   * top-k neighbour smoothing of the per-patch class probabilities on the frozen-trunk features
//     (criterions/seg_criterion.py:197-213): L2-normalise the features, cosine similarities (the NT GEMM of
//     gemm.hip, fp32 out), top-k per row, softmax of the logits, `iters` rounds of "mean over my k nearest patches";
//   * metrics at the ORIGINAL image resolution (:289-347): bilinear resize of the [hp, wp] score grid to [h, w]
//     (align_corners=False, any ratio), argmax, the three area histograms and the display cross entropy, without
//     materialising the [h*w, n] score tensor.
// All of it is HBM/latency-bound integer-and-elementwise work on a few MB: plain one-thread-per-element kernels,
// 16-byte accesses where rows allow it, integer atomics only (order-independent => deterministic).
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

// out[r, :] = x[r, :] / max(||x[r, :]||_2, 1e-12)   (F.normalize), one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const bf16_t* x, bf16_t* out, int rows, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xp = x + (long long)row * D;
  float ss = 0.f;
  for (int c = lane * 8; c < D; c += 512) {
    U128 u; u.v = *reinterpret_cast<const uint4*>(xp + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = bflo(u.w[e]), b = bfhi(u.w[e]); ss += a * a + b * b; }
  }
  const float inv = 1.f / fmaxf(sqrtf(warp_sum(ss)), 1e-12f);
  for (int c = lane * 8; c < D; c += 512) {
    U128 u; u.v = *reinterpret_cast<const uint4*>(xp + c);
    U128 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.w[e] = pack2bf(bflo(u.w[e]) * inv, bfhi(u.w[e]) * inv);
    *reinterpret_cast<uint4*>(out + (long long)row * D + c) = o.v;
  }
}

// idx[r, 0..k) = indices of the k largest entries of sim[r, 0..N) (ties: lower index first), one wave per row
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* sim, int* idx, int rows, int N, int k) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sp = sim + (long long)row * N;
  int won[8];
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < N; c += 64) {
      bool taken = false;
      for (int t = 0; t < j; ++t) taken |= (won[t] == c);
      const float v = sp[c];
      if (!taken && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    won[j] = bi;
    if (lane == 0) idx[(long long)row * k + j] = bi;
  }
}

// prob[r, 0..n) = softmax(logits[r, 0..n) * inv_t) in fp32 (do_softmax) or just the fp32 copy; one thread per row
__global__ void softmax_rows_kernel(const bf16_t* logits, long long row_bs, int rows_per_batch, int ld, float* prob, int rows,
                                    int n, float inv_t, int do_softmax) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const bf16_t* lp = logits + (long long)(r / rows_per_batch) * row_bs + (long long)(r % rows_per_batch) * ld;
  float* pp = prob + (long long)r * n;
  if (!do_softmax) {
    for (int c = 0; c < n; ++c) pp[c] = bf2f(lp[c]);
    return;
  }
  float m = -INFINITY;
  for (int c = 0; c < n; ++c) m = fmaxf(m, bf2f(lp[c]) * inv_t);
  float s = 0.f;
  for (int c = 0; c < n; ++c) { const float e = __expf(bf2f(lp[c]) * inv_t - m); pp[c] = e; s += e; }
  const float inv = 1.f / s;
  for (int c = 0; c < n; ++c) pp[c] *= inv;
}

// out[b, p, c] = mean_j in[b, idx[b, p, j], c]
__global__ void gather_mean_kernel(const float* in, const int* idx, float* out, int B, int P, int n, int k) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)B * P * n) return;
  const int c = (int)(gid % n);
  const long long bp = gid / n;
  const int b = (int)(bp / P);
  float s = 0.f;
  for (int j = 0; j < k; ++j) s += in[((long long)b * P + idx[bp * k + j]) * n + c];
  out[gid] = s / (float)k;
}

// metrics at the original resolution: one thread per pixel of the [h, w] label map
__global__ __launch_bounds__(256) void seg_eval_kernel(const float* scores, int hp, int wp, int n, const long long* target,
                                                       int h, int w, long long seg0, unsigned long long* hist /*[3][n]*/,
                                                       float* loss_part /*[nblk][2]*/) {
  __shared__ float red[2][4];
  const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (pix < (long long)h * w) {
    const long long t = target[pix] - seg0;
    if (t >= 0 && t < n) {                       // pad / eos / ignore (= n) are outside [0, n)
      const int y = (int)(pix / w), x = (int)(pix % w);
      // F.interpolate(bilinear, align_corners=False): src = (dst + 0.5) * in/out - 0.5, clamped at 0
      const float sy = fmaxf(((float)y + 0.5f) * ((float)hp / (float)h) - 0.5f, 0.f);
      const float sx = fmaxf(((float)x + 0.5f) * ((float)wp / (float)w) - 0.5f, 0.f);
      const int y0 = min((int)sy, hp - 1), x0 = min((int)sx, wp - 1);
      const int y1 = min(y0 + 1, hp - 1), x1 = min(x0 + 1, wp - 1);
      const float ly = sy - (float)y0, lx = sx - (float)x0;
      const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
      const float* p00 = scores + ((long long)y0 * wp + x0) * n;
      const float* p01 = scores + ((long long)y0 * wp + x1) * n;
      const float* p10 = scores + ((long long)y1 * wp + x0) * n;
      const float* p11 = scores + ((long long)y1 * wp + x1) * n;
      float best = -INFINITY, m = -INFINITY, s = 0.f, vt = 0.f;
      int arg = 0;
      for (int c = 0; c < n; ++c) {
        const float v = w00 * p00[c] + w01 * p01[c] + w10 * p10[c] + w11 * p11[c];
        if (v > best) { best = v; arg = c; }     // first maximum, like torch.argmax on ties
        if (v > m) { s = s * __expf(m - v) + 1.f; m = v; } else s += __expf(v - m);
        if (c == (int)t) vt = v;
      }
      loss = m + __logf(s) - vt;
      cnt = 1.f;
      atomicAdd(&hist[n + arg], 1ull);           // predicted-label area
      atomicAdd(&hist[2 * n + (int)t], 1ull);     // label area
      if (arg == (int)t) atomicAdd(&hist[arg], 1ull);   // intersection
    }
  }
  loss = warp_sum(loss); cnt = warp_sum(cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_part[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    loss_part[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

}  // namespace

extern "C" int ifseg_l2norm_rows_bf16(const void* x, void* out, int rows, int D, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if (D & 7) return IFSEG_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, rows, D);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_topk_rows_f32(const float* sim, int* idx, int rows, int N, int k, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if (k < 1 || k > 8 || k > N) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(topk_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, sim, idx, rows, N, k);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_softmax_rows(const void* logits, long long batch_stride, int rows_per_batch, int ld, float* prob,
                                  int rows, int n, float inv_temperature, int do_softmax, void* stream) {
  (void)hipGetLastError();
  if (rows <= 0) return 0;
  if (n <= 0 || rows_per_batch <= 0) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const bf16_t*)logits,
                     batch_stride, rows_per_batch, ld, prob, rows, n, inv_temperature, do_softmax);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_gather_mean(const float* in, const int* idx, float* out, int B, int P, int n, int k, void* stream) {
  (void)hipGetLastError();
  const long long total = (long long)B * P * n;
  if (total <= 0) return 0;
  if (k < 1) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(gather_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, idx, out,
                     B, P, n, k);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_seg_eval(const float* scores, int hp, int wp, int n, const long long* target, int h, int w,
                              long long seg_id_offset, unsigned long long* hist, float* loss_part, int nblocks,
                              void* stream) {
  (void)hipGetLastError();
  const long long npix = (long long)h * w;
  if (npix <= 0) return 0;
  if (n <= 0 || hp <= 0 || wp <= 0 || nblocks != (int)((npix + 255) / 256)) return IFSEG_ERR_BAD_ARG;
  hipLaunchKernelGGL(seg_eval_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, scores, hp, wp, n, target, h, w,
                     seg_id_offset, hist, loss_part);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
