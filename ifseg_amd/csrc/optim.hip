// Flat-arena optimizer kernels (gfx950): global grad-norm and a fused
// clip + Adam(W) + bf16 write-back, one launch over the whole parameter arena.
// Reference semantics: trainer.py:865-907 (multiply_grads, clip_grad_norm 1.0,
// optimizer.step) with fairseq/optim/adam.py:158-240 (decoupled weight decay,
// bias-corrected step) and fp16_optimizer.py:96-222 (fp32 master copy).
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_bf16_kernel(const bf16_t* g, long long n, float* part) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x * 8;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 7 < n) {
      uint4 u = *reinterpret_cast<const uint4*>(g + i);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = bflo(w[e]), b = bfhi(w[e]); acc += a * a + b * b; }
    } else {
      for (long long k = i; k < n; ++k) { const float a = bf2f(g[k]); acc += a * a; }
    }
  }
  acc = warp_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void finish_norm_kernel(const float* part, int nparts, float* out_sumsq) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
  acc = warp_sum(acc);
  if (threadIdx.x == 0) out_sumsq[0] = acc;
}

// p32, m, v: fp32 masters; g: bf16 grads; p16: bf16 model copy.  gscale multiplies
// every gradient (world/sample_size factor); the clip coefficient is derived on
// device from *sumsq (of the unscaled grads) so there is no host sync.
__global__ __launch_bounds__(256) void adam_kernel(float* p32, const bf16_t* g, float* m, float* v, bf16_t* p16,
                                                   long long n, float lr, float beta1, float beta2, float eps, float wd,
                                                   float bc1, float bc2, float gscale, float max_norm,
                                                   const float* sumsq, int* overflow, const float* hyper) {
  if (hyper) {      // per-update scalars from device memory (a captured step replays with the current schedule)
    lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; gscale = hyper[3];
  }
  float coef = gscale;
  if (sumsq && !isfinite(sumsq[0])) {
    // trainer.py:895-904: a NaN / Inf gradient norm must not reach the fp32 masters or the Adam moments
    // (fminf(1, NaN) would evaluate to 1 and apply the poisoned step): skip the update, raise the flag
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) overflow[0] = 1;
    return;
  }
  if (max_norm > 0.f && sumsq) {
    const float norm = sqrtf(sumsq[0]) * gscale;
    coef *= fminf(1.f, max_norm / (norm + 1e-6f));
  }
  const float step_size = lr * sqrtf(bc2) / bc1;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 P = *reinterpret_cast<float4*>(p32 + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
      uint2 gw = *reinterpret_cast<const uint2*>(g + i);
      float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
      const float gg[4] = {bflo(gw.x) * coef, bfhi(gw.x) * coef, bflo(gw.y) * coef, bfhi(gw.y) * coef};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mm[e] = mm[e] * beta1 + gg[e] * (1.f - beta1);
        vv[e] = vv[e] * beta2 + gg[e] * gg[e] * (1.f - beta2);
        const float denom = sqrtf(vv[e]) + eps;
        pp[e] = pp[e] - wd * lr * pp[e];
        pp[e] = pp[e] - step_size * mm[e] / denom;
      }
      *reinterpret_cast<float4*>(p32 + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      *reinterpret_cast<uint2*>(p16 + i) = make_uint2(pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]));
    } else {
      for (long long k = i; k < n; ++k) {
        const float gk = bf2f(g[k]) * coef;
        const float mk = m[k] * beta1 + gk * (1.f - beta1), vk = v[k] * beta2 + gk * gk * (1.f - beta2);
        float pk = p32[k];
        pk -= wd * lr * pk;
        pk -= step_size * mk / (sqrtf(vk) + eps);
        p32[k] = pk; m[k] = mk; v[k] = vk; p16[k] = f2bf(pk);
      }
    }
  }
}

}  // namespace

extern "C" int ifseg_grad_sumsq_bf16(const void* g, long long n, float* workspace /* >= 1024 floats */,
                                     float* out_sumsq, void* stream) {
  (void)hipGetLastError();
  const int nblk = 1024;
  hipLaunchKernelGGL(sumsq_bf16_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, n, workspace);
  hipLaunchKernelGGL(finish_norm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nblk, out_sumsq);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_adam_step(float* p32, const void* g, float* m, float* v, void* p16, long long n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                               float max_norm, const float* sumsq, int* overflow, const float* hyper, void* stream) {
  (void)hipGetLastError();
  if (n <= 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  // grid: two blocks per CU.  Sweep on MI355X over 109 M parameters (tools/adam_bench.py, profiles/round6_adam_grid.txt): 256 blocks
  // 789 us, 512: 573, 768: 578, 1024: 588, 1536: 618, 2048 (until round 6): 620, 4096: 620 -- 5.33 TB/s against a torch copy's 5.0
  hipLaunchKernelGGL(adam_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, p32, (const bf16_t*)g, m, v,
                     (bf16_t*)p16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, max_norm, sumsq, overflow, hyper);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
