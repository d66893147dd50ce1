// Variable-aspect evaluation (BASELINE configs[4], criterions/seg_criterion.py:194-217: batch 1, native aspect ratio, a
// feature grid (h, w) other than the trained (oh, oh)): the reference resizes its position tables and its
// [H, P0, P0] relative-position bias with two bilinear interpolations per layer (encoder_module.py:360-368,802-808,
// decoder_module.py:541-548,603-627).  Both resizes are linear and separable -- bias' = Wq . B0 . Wk^T with four-tap
// interpolation matrices -- but not translation invariant (fractional phases, edge clamping), so the resized bias is not a
// function of (i - j) any more and cannot be indexed arithmetically by the attention kernels: it is built ONCE per
// (layer, h, w) by the kernels below as a dense fp32 [H, T, S] tensor (the engine caches it: the weights are fixed at
// evaluation time and validation sets repeat a handful of aspect ratios; 288 GB of HBM hold dozens of shapes).
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

// taps of one resized grid position: four source positions (raster index on the original oh x ow grid) and weights
// (F.interpolate, bilinear, align_corners = False: src = max(0, (d + 0.5) * scale - 0.5), clamped neighbour)
struct TapGeo { int h, w, oh, ow; };

__device__ __forceinline__ void taps1d(int d, int dn, int sn, int& i0, int& i1, float& w1) {
  const float scale = (float)sn / (float)dn;
  float src = ((float)d + 0.5f) * scale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 > sn - 1 ? sn - 1 : i0;
  i1 = i0 + 1 > sn - 1 ? sn - 1 : i0 + 1;
  w1 = src - (float)i0;
}

__device__ __forceinline__ void taps2d(int p, const TapGeo& g, int (&y)[2], int (&x)[2], float (&wy)[2], float (&wx)[2]) {
  const int py = p / g.w, px = p - py * g.w;
  float a;
  taps1d(py, g.h, g.oh, y[0], y[1], a); wy[0] = 1.f - a; wy[1] = a;
  taps1d(px, g.w, g.ow, x[0], x[1], a); wx[0] = 1.f - a; wx[1] = a;
}

// out[hd][i][j] (fp32 [H, T, S], T = S = P + Lt, internal order [grid | tail]):
//   grid x grid : sum over the 4 x 4 taps of wq wk table2d[hd][code0(a) - code0(b) + code_bias0]   (a, b on the oh x ow grid)
//   tail x tail : rel1d[hd][(i - P) - (j - P) + Lt - 1]                       (encoder text block / decoder [0,0] corner)
//   grid x tail : relx[hd][0],  tail x grid : relx[hd][1]                    (decoder bos column / row; zero in the encoder)
// causal (decoder, reference order [bos, patches] = internal order with the tail FIRST): -inf where the key comes after
// the query -- the mask travels inside the dense bias because the kernels' causal tile skipping needs P % 64 == 0
__global__ __launch_bounds__(256) void resized_bias_kernel(float* __restrict__ out, const float* __restrict__ table2d,
                                                           const float* __restrict__ rel1d, const float* __restrict__ relx,
                                                           int H, int P, int Lt, TapGeo g, int n2d0, int causal, int ld) {
  extern __shared__ float sTab[];
  const int T = P + Lt;
  const int hd = blockIdx.z;
  for (int i = threadIdx.x; i < n2d0; i += 256) sTab[i] = table2d[(long long)hd * n2d0 + i];
  __syncthreads();
  const int i = blockIdx.y;                           // query row
  const int stride0 = 2 * g.ow - 1, bias0 = (g.oh - 1) * stride0 + (g.ow - 1);
  float* orow = out + ((long long)hd * T + i) * ld;
  if (i < P) {
    int ya[2], xa[2]; float wya[2], wxa[2];
    taps2d(i, g, ya, xa, wya, wxa);
    int ca[4]; float wa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ca[t] = ya[t >> 1] * stride0 + xa[t & 1] + bias0; wa[t] = wya[t >> 1] * wxa[t & 1]; }
    for (int j = blockIdx.x * 256 + threadIdx.x; j < T; j += gridDim.x * 256) {
      float v;
      if (j < P) {
        int yb[2], xb[2]; float wyb[2], wxb[2];
        taps2d(j, g, yb, xb, wyb, wxb);
        // the reference resizes along the keys first, then along the queries: inner sum over the key taps
        v = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float inner = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int cb = yb[u >> 1] * stride0 + xb[u & 1];
            inner += (wyb[u >> 1] * wxb[u & 1]) * sTab[ca[t] - cb];
          }
          v += wa[t] * inner;
        }
      } else {
        v = relx ? relx[hd * 2 + 0] : 0.f;
      }
      if (causal && j < P && j > i) v = -INFINITY;          // a grid key after the query (the tail = bos stays visible)
      orow[j] = v;
    }
  } else {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < T; j += gridDim.x * 256) {
      float v;
      if (j < P) v = causal ? -INFINITY : (relx ? relx[hd * 2 + 1] : 0.f);   // bos (reference position 0) sees itself only
      else v = (causal && j > i) ? -INFINITY : (rel1d ? rel1d[(long long)hd * (2 * Lt - 1) + (i - j) + Lt - 1] : 0.f);
      orow[j] = v;
    }
  }
}

// dst[p][c] (bf16 [h*w, C]) = bilinear resize of src[(y * src_stride + x + src_off)][c] over the oh x ow grid, interpolated
// in fp32 and rounded once (the reference: table rows -> float -> F.interpolate -> bf16)
__global__ __launch_bounds__(256) void resize_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int C,
                                                          TapGeo g, int src_stride, int src_off, int ld_src) {
  const int p = blockIdx.x;
  int y[2], x[2]; float wy[2], wx[2];
  taps2d(p, g, y, x, wy, wx);
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      v += (wy[t >> 1] * wx[t & 1]) * bf2f(src[(long long)(y[t >> 1] * src_stride + x[t & 1] + src_off) * ld_src + c]);
    dst[(long long)p * C + c] = f2bf(v);
  }
}

}  // namespace

extern "C" int ifseg_resized_rel_bias(float* out, const float* table2d, const float* rel1d, const float* relx, int H, int h,
                                      int w, int oh, int ow, int Lt, int causal, int ld, void* stream) {
  (void)hipGetLastError();
  if (!out || !table2d || H <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || Lt < 0) return IFSEG_ERR_BAD_ARG;
  const int P = h * w, T = P + Lt, n2d0 = (2 * oh - 1) * (2 * ow - 1);
  if (ld <= 0) ld = T;
  if (ld < T) return IFSEG_ERR_BAD_ARG;
  const size_t lds = (size_t)n2d0 * sizeof(float);
  if (lds > 160 * 1024) return IFSEG_ERR_BAD_SHAPE;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)resized_bias_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  TapGeo g{h, w, oh, ow};
  const int bx = (T + 255) / 256 < 8 ? (T + 255) / 256 : 8;
  hipLaunchKernelGGL(resized_bias_kernel, dim3(bx, T, H), dim3(256), lds, (hipStream_t)stream, out, table2d, rel1d, relx, H, P,
                     Lt, g, n2d0, causal, ld);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_resize_rows_bilinear(const void* src, void* dst, int C, int h, int w, int oh, int ow, int src_stride,
                                          int src_off, int ld_src, void* stream) {
  (void)hipGetLastError();
  if (!src || !dst || C <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return IFSEG_ERR_BAD_ARG;
  TapGeo g{h, w, oh, ow};
  hipLaunchKernelGGL(resize_rows_kernel, dim3(h * w), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, C, g,
                     src_stride, src_off, ld_src);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
