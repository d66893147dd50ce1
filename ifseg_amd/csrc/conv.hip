// ResNet stem for the frozen patch-embed trunk (gfx950): 7x7/2 conv (3->64) with
// folded FrozenBN + ReLU, and the 3x3/2 max-pool.  The remaining 1x1 / 3x3 convs go
// through the implicit-GEMM MFMA path in gemm.hip.
// Reference: ResNet._forward_impl resnet.py:215-220 (conv1, bn1, relu, maxpool).
#include "common.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr int ST_TH = 8, ST_TW = 32;                 // output tile per block (256 pixels)
constexpr int ST_PH = (ST_TH - 1) * 2 + 7, ST_PW = (ST_TW - 1) * 2 + 7;  // 21 x 69 input patch

// in: NHWC bf16 with C padded to 4; w: fp32 [7*7*3][64] (BN scale folded); shift fp32 [64]
__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_t* in, const float* w, const float* shift,
                                                        bf16_t* out, int B, int H, int W, int OH, int OW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sW = reinterpret_cast<float*>(smem);                 // 147*64
  float* sIn = sW + 147 * 64;                                 // ST_PH*ST_PW*3
  const int tid = threadIdx.x;
  const int tiles_x = (OW + ST_TW - 1) / ST_TW, tiles_y = (OH + ST_TH - 1) / ST_TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x % (tiles_x * tiles_y);
  const int oy0 = (t / tiles_x) * ST_TH, ox0 = (t % tiles_x) * ST_TW;
  for (int i = tid; i < 147 * 64; i += 256) sW[i] = w[i];
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  for (int i = tid; i < ST_PH * ST_PW; i += 256) {
    const int py = i / ST_PW, px = i % ST_PW, iy = iy0 + py, ix = ix0 + px;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      uint2 u = *reinterpret_cast<const uint2*>(in + (((long long)b * H + iy) * W + ix) * 4);
      v0 = bflo(u.x); v1 = bfhi(u.x); v2 = bflo(u.y);
    }
    sIn[i * 3 + 0] = v0; sIn[i * 3 + 1] = v1; sIn[i * 3 + 2] = v2;
  }
  __syncthreads();
  const int ty = tid / ST_TW, tx = tid % ST_TW;
  float acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    for (int kx = 0; kx < 7; ++kx) {
      const float* ip = sIn + ((ty * 2 + ky) * ST_PW + tx * 2 + kx) * 3;
      const float* wp = sW + (ky * 7 + kx) * 3 * 64;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = ip[ci];
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
          const float4 w4 = *reinterpret_cast<const float4*>(wp + ci * 64 + c);
          acc[c] += v * w4.x; acc[c + 1] += v * w4.y; acc[c + 2] += v * w4.z; acc[c + 3] += v * w4.w;
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < OH && ox < OW) {
    bf16_t* op = out + (((long long)b * OH + oy) * OW + ox) * 64;
#pragma unroll
    for (int c = 0; c < 64; c += 8) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaxf(acc[c + e] + shift[c + e], 0.f);
      *reinterpret_cast<uint4*>(op + c) =
          make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    }
  }
}

// The same convolution on the matrix cores (round 4: the direct kernel above ran at 46 TFLOP/s, 424 us for 16 images of
// 512 x 512 -- ten times its 40 us HBM bound: 134 MB of output).  Implicit GEMM with the filter row padded to 8 taps and the
// channels to 4: K = 7 x 8 x 4 = 224 = 14 MFMA k-steps, k = ky*32 + kx*4 + ci (taps kx = 7 and channel 3 carry zero
// weights).  Computed swapped, C^T[channel][pixel] = W[channel][k] . patch[k][pixel]: a lane is an output pixel holding runs of
// 4 consecutive channels, and the B fragment of a k-step -- 8 consecutive k = two neighbouring input pixels x 4 channels -- is
// ONE aligned 16-byte LDS read of the raw NHWC(4) patch (no im2col).  The 28 weight fragments of a wave (2 channel tiles x
// 14 k-steps) stay in registers; workgroups walk output tiles of 8 x 32 pixels.
// Weights as THREE bf16 terms (wt: bf16 [3][64][224], t0 = bf16(w), t1 = bf16(w - t0), t2 = bf16(w - t0 - t1): 24 mantissa bits,
// the fp32 weight exactly): the direct kernel kept them in fp32; with plain bf16 weights the evaluation fixture's
// neighbour-smoothed histograms moved three times further from the reference's (the first layer's rounding reaches every
// feature), and with two terms 0.25 % of the stem's outputs still rounded the other way -- enough to move the plain argmax
// agreement of the 150-class golden (a statistic of near-ties) from 0.9775 to 0.9688.  Terms 1 and 2 sit in LDS (row pitch 464
// bytes: conflict-free 16-byte reads by 32 consecutive channels); three times the MFMAs of a kernel that is bound by its 134 MB
// of output.
constexpr int STM_PW = 72;                            // patch row pitch in pixels (>= 2*31 + 8)
constexpr int STM_WP = 464;                           // lo-weight row pitch in bytes (224 bf16 = 448, padded)
__global__ __launch_bounds__(256) void stem_conv_mfma_kernel(const bf16_t* in, const bf16_t* wt, const float* shift,
                                                             bf16_t* out, int B, int H, int W, int OH, int OW) {
  __shared__ __attribute__((aligned(16))) uint2 sIn[ST_PH * STM_PW];       // 21 x 72 pixels x 8 bytes = 12 KiB
  __shared__ __attribute__((aligned(16))) unsigned char sWl[2 * 64 * STM_WP];  // terms 1, 2: [2][64][224] bf16, 58 KiB
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, tx = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16x8 wf[2][14];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) {
      U128 u; u.v = *reinterpret_cast<const uint4*>(wt + (ct * 32 + tx) * 224 + ks * 16 + half * 8);
      wf[ct][ks] = u.b;
    }
  for (int i = tid; i < 2 * 64 * 28; i += 256) {      // 28 16-byte chunks per channel row
    const int c = i / 28, ch = i - c * 28;             // c = term * 64 + channel
    *reinterpret_cast<uint4*>(sWl + c * STM_WP + ch * 16) = *reinterpret_cast<const uint4*>(wt + 64 * 224 + c * 224 + ch * 8);
  }
  float sh[2][4][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 s4 = *reinterpret_cast<const float4*>(shift + ct * 32 + 8 * rg + 4 * half);
      sh[ct][rg][0] = s4.x; sh[ct][rg][1] = s4.y; sh[ct][rg][2] = s4.z; sh[ct][rg][3] = s4.w;
    }
  const int tiles_x = (OW + ST_TW - 1) / ST_TW, tiles_y = (OH + ST_TH - 1) / ST_TH;
  const int total = B * tiles_x * tiles_y;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int b = w / (tiles_x * tiles_y), t = w % (tiles_x * tiles_y);
    const int oy0 = (t / tiles_x) * ST_TH, ox0 = (t % tiles_x) * ST_TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    __syncthreads();                                  // the previous tile's reads are done
    for (int i = tid; i < ST_PH * STM_PW; i += 256) {
      const int py = i / STM_PW, px = i - py * STM_PW, iy = iy0 + py, ix = ix0 + px;
      uint2 u = make_uint2(0, 0);
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) u = *reinterpret_cast<const uint2*>(in + (((long long)b * H + iy) * W + ix) * 4);
      sIn[i] = u;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][ct][e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ty = wave * 2 + i;
          const bf16x8 pf = lds_read_b128(reinterpret_cast<const unsigned char*>(sIn + (2 * ty + ky) * STM_PW + 2 * tx + 4 * g + 2 * half));
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ct][ky * 2 + g], pf, acc[i][ct], 0, 0, 0);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              const bf16x8 wl = lds_read_b128(sWl + (tm * 64 + ct * 32 + tx) * STM_WP + (ky * 2 + g) * 32 + half * 16);
              acc[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, pf, acc[i][ct], 0, 0, 0);
            }
        }
      }
    // lane = pixel (oy0 + 2 wave + i, ox0 + tx); element r <-> channel ct*32 + 8*(r>>2) + 4*half + (r&3)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = oy0 + wave * 2 + i, ox = ox0 + tx;
      if (oy >= OH || ox >= OW) continue;
      bf16_t* op = out + (((long long)b * OH + oy) * OW + ox) * 64 + 4 * half;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc[i][ct][rg * 4 + e] + sh[ct][rg][e], 0.f);
          *reinterpret_cast<uint2*>(op + ct * 32 + 8 * rg) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
    }
  }
}

// 3x3 stride 2 pad 1 max-pool on NHWC bf16, 8 channels per thread
__global__ void maxpool3x3s2_kernel(const bf16_t* in, bf16_t* out, int B, int H, int W, int C, int OH, int OW) {
  const int nch = C >> 3;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * OH * OW * nch;
  if (gid >= total) return;
  const int c = (int)(gid % nch);
  const long long p = gid / nch;
  const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), b = (int)(p / ((long long)OW * OH));
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      uint4 u = *reinterpret_cast<const uint4*>(in + (((long long)b * H + iy) * W + ix) * C + c * 8);
      m[0] = fmaxf(m[0], bflo(u.x)); m[1] = fmaxf(m[1], bfhi(u.x)); m[2] = fmaxf(m[2], bflo(u.y));
      m[3] = fmaxf(m[3], bfhi(u.y)); m[4] = fmaxf(m[4], bflo(u.z)); m[5] = fmaxf(m[5], bfhi(u.z));
      m[6] = fmaxf(m[6], bflo(u.w)); m[7] = fmaxf(m[7], bfhi(u.w));
    }
  }
  *reinterpret_cast<uint4*>(out + p * C + c * 8) =
      make_uint4(pack2bf(m[0], m[1]), pack2bf(m[2], m[3]), pack2bf(m[4], m[5]), pack2bf(m[6], m[7]));
}

}  // namespace

extern "C" int ifseg_stem_conv7x7(const void* in_nhwc4, const float* w, const float* shift, void* out, int B, int H,
                                  int W, void* stream) {
  (void)hipGetLastError();
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const int tiles = ((OH + ST_TH - 1) / ST_TH) * ((OW + ST_TW - 1) / ST_TW);
  const size_t lds = (147 * 64 + ST_PH * ST_PW * 3) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)stem_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(stem_conv_kernel, dim3(B * tiles), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)in_nhwc4,
                     w, shift, (bf16_t*)out, B, H, W, OH, OW);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_stem_conv7x7_mfma(const void* in_nhwc4, const void* wt, const float* shift, void* out, int B, int H,
                                       int W, void* stream) {
  (void)hipGetLastError();
  if (!in_nhwc4 || !wt || !shift || !out || B <= 0 || H <= 0 || W <= 0 || (((size_t)wt | (size_t)shift | (size_t)out) & 15)) return IFSEG_ERR_BAD_ARG;
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const int total = B * ((OH + ST_TH - 1) / ST_TH) * ((OW + ST_TW - 1) / ST_TW);
  const int grid = total < 1024 ? total : 1024;
  hipLaunchKernelGGL(stem_conv_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_nhwc4,
                     (const bf16_t*)wt, shift, (bf16_t*)out, B, H, W, OH, OW);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, void* stream) {
  (void)hipGetLastError();
  if (C & 7) return IFSEG_ERR_BAD_SHAPE;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)B * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, (bf16_t*)out, B, H, W, C, OH, OW);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
