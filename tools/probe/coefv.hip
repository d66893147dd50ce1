// variants of ffn_ln_coef_kernel for the concurrency-corruption hunt
#include <hip/hip_runtime.h>
#include "../../ifseg_amd/csrc/common.h"
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
struct Ptrs { const bf16_t* w2[32]; const float* gamma[32]; const float* beta[32]; const bf16_t* b2[32]; float* coef[32]; };
template <int V>
__global__ __launch_bounds__(256) void coefv_kernel(Ptrs pt, int ldw, int J, int N) {
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
    if (V >= 10 && V <= 19) {
      // the SLP code's FIRST step in explicit form: (sa, sb) += (g.x w0 + g.y w1, b.x w0 + b.y w1) built from a half-swapping
      // v_pk_mov_b32 and a cross-selecting v_pk_mul_f32; the other six terms as plain packed FMAs
      //   V10: pk_mov op_sel + pk_mul cross op_sel (as clang emits)   V11: two v_mov instead of the pk_mov
      //   V12: pk_mov kept, products by two scalar multiplies          V13: everything plain (control)
      typedef float f2v __attribute__((ext_vector_type(2)));
      const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      unpack8(*reinterpret_cast<const uint4*>(row + k), w);
      f2v acc = {sa, sb};
      f2v gxy = {gs[0], gs[1]}, bxy = {bs[0], bs[1]}, w01 = {w[0], w[1]}, mv, t, a0 = {gs[0], bs[1]};
      if (V == 10 || V == 12 || V >= 14) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(mv) : "v"(gxy), "v"(bxy));      // (g.y, b.x)
      else { mv.x = gs[1]; mv.y = bs[0]; asm volatile("" : "+v"(mv)); }
      if (V == 10 || V == 11) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(mv), "v"(w01));   // (g.y w1, b.x w0)
      else if (V == 18) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(w01), "v"(mv));   // the cross read on src0
      else if (V == 19) { f2v w10 = {w[1], w[0]}; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(mv), "v"(w10)); }           // plain, pre-swapped pair
      else if (V == 14) asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(mv), "v"(w01));
      else if (V == 15) asm volatile("s_nop 1\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(mv), "v"(w01));
      else if (V == 16) asm volatile("s_nop 3\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(mv), "v"(w01));
      else if (V == 17) asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(mv), "v"(w01));
      else { t.x = mv.x * w[1]; t.y = mv.y * w[0]; asm volatile("" : "+v"(t)); }
      if (V == 13) { t.x += a0.x * w[0]; t.y += a0.y * w[1]; asm volatile("" : "+v"(t)); }
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(a0), "v"(w01));                                              // + (g.x w0, b.y w1)
#pragma unroll
      for (int e = 2; e < 8; ++e) {
        f2v gb = {gs[e], bs[e]}, ww;
        ww.x = w[e];
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(t) : "v"(gb), "v"(ww));
      }
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(t));
      sa = acc.x; sb = acc.y;
      continue;
    }
    if (V == 7 || V == 8 || V == 9) {
      // explicit packed math: acc = {sa, sb}; V7: src1 = {w, w} fully defined; V8: same but accumulators parked away from v[0:1]
      typedef float f2v __attribute__((ext_vector_type(2)));
      const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      unpack8(*reinterpret_cast<const uint4*>(row + k), w);
      f2v acc = {sa, sb};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f2v gb = {gs[e], bs[e]}, ww;
        ww.x = w[e];
        if (V != 8) ww.y = w[e];                  // V8: the unselected half is left undefined
        if (V == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(gb), "v"(ww));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(gb), "v"(ww));
      }
      sa = acc.x; sb = acc.y;
      continue;
    }
    if (V == 5 || V == 6) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      u4v wr = *reinterpret_cast<const u4v*>(row + k);
      f4v G0 = *reinterpret_cast<const f4v*>(gamma + k), G1 = *reinterpret_cast<const f4v*>(gamma + k + 4);
      f4v B0 = *reinterpret_cast<const f4v*>(beta + k), B1 = *reinterpret_cast<const f4v*>(beta + k + 4);
      if (V == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr), "+v"(G0), "+v"(G1), "+v"(B0), "+v"(B1));
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(wr), "+v"(G0), "+v"(G1), "+v"(B0), "+v"(B1));
      unpack8(make_uint4(wr.x, wr.y, wr.z, wr.w), w);
      sa += w[0] * G0.x + w[1] * G0.y + w[2] * G0.z + w[3] * G0.w + w[4] * G1.x + w[5] * G1.y + w[6] * G1.z + w[7] * G1.w;
      sb += w[0] * B0.x + w[1] * B0.y + w[2] * B0.z + w[3] * B0.w + w[4] * B1.x + w[5] * B1.y + w[6] * B1.z + w[7] * B1.w;
      continue;
    }
    sa += w[0] * g0.x + w[1] * g0.y + w[2] * g0.z + w[3] * g0.w + w[4] * g1.x + w[5] * g1.y + w[6] * g1.z + w[7] * g1.w;
    if (V == 2) asm volatile("" : "+v"(sa));
    sb += w[0] * b0.x + w[1] * b0.y + w[2] * b0.z + w[3] * b0.w + w[4] * b1.x + w[5] * b1.y + w[6] * b1.z + w[7] * b1.w;
    if (V == 2) asm volatile("" : "+v"(sb));
  }
  if (V == 1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
  } else if (V == 3) {
    sb = warp_sum(sb); sa = warp_sum(sa);
  } else if (V == 4) {
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    sa = warp_sum(sa);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    sb = warp_sum(sb);
  } else {
    sa = warp_sum(sa); sb = warp_sum(sb);
  }
  if (lane == 0) { coef[j] = sa; coef[J + j] = sb + (b2 ? bf2f(b2[j]) : 0.f); }
}
extern "C" int coefv_launch(int variant, const void* const* w2, int ldw, const float* const* gamma, const float* const* beta,
                            const void* const* b2, float* const* coef, int L, int J, int N, void* stream) {
  Ptrs pt{};
  for (int l = 0; l < L; ++l) { pt.w2[l] = (const bf16_t*)w2[l]; pt.gamma[l] = gamma[l]; pt.beta[l] = beta[l]; pt.b2[l] = (const bf16_t*)b2[l]; pt.coef[l] = coef[l]; }
  dim3 g((J + 3) / 4, L), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL(coefv_kernel<0>, g, b, 0, s, pt, ldw, J, N); break;
    case 1: hipLaunchKernelGGL(coefv_kernel<1>, g, b, 0, s, pt, ldw, J, N); break;
    case 2: hipLaunchKernelGGL(coefv_kernel<2>, g, b, 0, s, pt, ldw, J, N); break;
    case 3: hipLaunchKernelGGL(coefv_kernel<3>, g, b, 0, s, pt, ldw, J, N); break;
    case 4: hipLaunchKernelGGL(coefv_kernel<4>, g, b, 0, s, pt, ldw, J, N); break;
    case 5: hipLaunchKernelGGL(coefv_kernel<5>, g, b, 0, s, pt, ldw, J, N); break;
    case 6: hipLaunchKernelGGL(coefv_kernel<6>, g, b, 0, s, pt, ldw, J, N); break;
    case 7: hipLaunchKernelGGL(coefv_kernel<7>, g, b, 0, s, pt, ldw, J, N); break;
    case 8: hipLaunchKernelGGL(coefv_kernel<8>, g, b, 0, s, pt, ldw, J, N); break;
    case 9: hipLaunchKernelGGL(coefv_kernel<9>, g, b, 0, s, pt, ldw, J, N); break;
    case 10: hipLaunchKernelGGL(coefv_kernel<10>, g, b, 0, s, pt, ldw, J, N); break;
    case 11: hipLaunchKernelGGL(coefv_kernel<11>, g, b, 0, s, pt, ldw, J, N); break;
    case 12: hipLaunchKernelGGL(coefv_kernel<12>, g, b, 0, s, pt, ldw, J, N); break;
    case 13: hipLaunchKernelGGL(coefv_kernel<13>, g, b, 0, s, pt, ldw, J, N); break;
    case 14: hipLaunchKernelGGL(coefv_kernel<14>, g, b, 0, s, pt, ldw, J, N); break;
    case 15: hipLaunchKernelGGL(coefv_kernel<15>, g, b, 0, s, pt, ldw, J, N); break;
    case 16: hipLaunchKernelGGL(coefv_kernel<16>, g, b, 0, s, pt, ldw, J, N); break;
    case 17: hipLaunchKernelGGL(coefv_kernel<17>, g, b, 0, s, pt, ldw, J, N); break;
    case 18: hipLaunchKernelGGL(coefv_kernel<18>, g, b, 0, s, pt, ldw, J, N); break;
    case 19: hipLaunchKernelGGL(coefv_kernel<19>, g, b, 0, s, pt, ldw, J, N); break;
  }
  return (int)hipGetLastError();
}
