// Fused criterion kernel for the segmentation loss (gfx950, HBM-bound):
//   bilinear upsample (align_corners=False) of per-patch logits [B, hp*wp, nseg] to pixel
//   resolution, masked mean cross entropy, its gradient w.r.t. the per-patch logits, argmax
//   and the per-class intersect / pred / label area histograms -- without ever materialising
//   the [B, H*W, nseg] fp32 score tensor (1.26 GB at B=8, 150 classes).
// Reference: criterions/seg_criterion.py:237-244 (upsample_logits), :269-347 (compute_loss),
//            :349-362 (compute_metric).  Scale factor H/hp == W/wp == 16 (every shipped config).
//
// One workgroup per low-res cell = one 16x16 pixel tile.  The tile reads the 3x3 cells its
// pixels interpolate from, each thread owns one pixel (online softmax over the classes), and
// the adjoint of the interpolation is applied separably (over x, then over y) in LDS, giving
// a [3][3][nseg] partial per tile; a second kernel gathers the <=9 partials of every cell.
// No global atomics: deterministic.
#include "common.h"
#include "prof.h"
#include "../../include/ifseg_hip.h"

namespace {

constexpr int TS = 16;       // pixels per cell side
constexpr int CH = 16;       // classes per gradient chunk
constexpr int NS_MAX = 512;  // max classes (LDS: 9 x (nseg | 1) floats + 3 x nseg ints, sized per launch)

__global__ __launch_bounds__(256) void seg_loss_tile_kernel(const bf16_t* logits, int ldl, long long lbs,
                                                            const long long* target, long long tbs, int hp, int wp,
                                                            int W, int nseg, long long seg0, long long pad_id,
                                                            long long eos_id, float* tile_partial, float* stats_part,
                                                            int* bad_label, float eps) {
  extern __shared__ float dyn[];
  const int ls = nseg | 1;                   // row stride of the 3 x 3 cells' logits: odd, so the four taps of a pixel
  float* sLog = dyn;                         // (rows k apart) fall into different banks
  int* sHist = reinterpret_cast<int*>(dyn + 9 * ls);      // [3][nseg]
  __shared__ float sWY[TS][3], sWX[TS][3];
  __shared__ float sD[256][CH + 1];
  __shared__ float sE[TS][3][CH];
  __shared__ float sRed[8];
  const int tid = threadIdx.x;
  const int ncell = hp * wp;
  const int b = blockIdx.x / ncell, cell = blockIdx.x % ncell;
  const int cy = cell / wp, cx = cell % wp;
  const int nstat = 2 + 3 * nseg;

  for (int i = tid; i < 9 * nseg; i += 256) {
    const int k = i / nseg, c = i % nseg;
    const int ry = cy - 1 + k / 3, rx = cx - 1 + k % 3;
    float v = 0.f;
    if (ry >= 0 && ry < hp && rx >= 0 && rx < wp) v = bf2f(logits[b * lbs + (long long)(ry * wp + rx) * ldl + c]);
    sLog[k * ls + c] = v;
  }
  for (int i = tid; i < 3 * nseg; i += 256) sHist[i] = 0;
  if (tid < 2 * TS) {
    // source index of F.interpolate(mode='bilinear', align_corners=False): max((dst+0.5)/s-0.5, 0)
    const bool isx = tid >= TS;
    const int t = tid & (TS - 1);
    const int c0 = isx ? cx : cy, lim = isx ? wp : hp;
    float src = ((c0 * TS + t) + 0.5f) * (1.f / TS) - 0.5f;
    src = fmaxf(src, 0.f);
    const int i0 = (int)src;
    const int i1 = min(i0 + 1, lim - 1);
    const float l1 = src - i0, l0 = 1.f - l1;
    const int s0 = i0 - (c0 - 1), s1 = i1 - (c0 - 1);
    float* wrow = isx ? sWX[t] : sWY[t];
#pragma unroll
    for (int k = 0; k < 3; ++k) wrow[k] = (k == s0 ? l0 : 0.f) + (k == s1 ? l1 : 0.f);
  }
  __syncthreads();

  const int py = tid >> 4, px = tid & 15;
  const float wy0 = sWY[py][0], wy1 = sWY[py][1], wy2 = sWY[py][2];
  const float wx0 = sWX[px][0], wx1 = sWX[px][1], wx2 = sWX[px][2];
  // A pixel's bilinear value has at most 2 x 2 non-zero taps among the 3 x 3 cells around its own: rows {0,1} or {1,2},
  // columns likewise.  Only those four are read and accumulated, in the same (increasing k) order as the nine-term sum this
  // replaces -- whose other five terms were exact zeros: the result is bit-identical, at 4 instead of 9 LDS reads per class
  // (the loop over the classes runs twice per pixel; at 150 - 171 classes this kernel was 0.6 - 1.0 ms of a step).
  const int kyl = (wy0 != 0.f) ? 0 : 1, kxl = (wx0 != 0.f) ? 0 : 1;
  const float wya = kyl ? wy1 : wy0, wyb = kyl ? wy2 : wy1, wxa = kxl ? wx1 : wx0, wxb = kxl ? wx2 : wx1;
  const float wA = wya * wxa, wB = wya * wxb, wC = wyb * wxa, wD = wyb * wxb;
  const float* lg = sLog + (kyl * 3 + kxl) * ls;
  auto value = [&](int c) {
    float v = 0.f;
    v += wA * lg[c];
    v += wB * lg[ls + c];
    v += wC * lg[3 * ls + c];
    v += wD * lg[4 * ls + c];
    return v;
  };
  const long long tg = target[b * tbs + (long long)(cy * TS + py) * W + (cx * TS + px)];
  // labels outside [seg0, seg0 + nseg) that are not pad / eos / ignore would index the LDS histograms out of bounds:
  // they are dropped here and reported by the criterion's (deferred) range check -- F.cross_entropy raises on them
  const bool valid = !(tg == pad_id || tg == eos_id || tg == seg0 + nseg) && tg >= seg0 && tg < seg0 + nseg;
  const int label = valid ? (int)(tg - seg0) : 0;
  // (rare) report a label that is neither a class nor pad / eos / ignore: the criterion raises on the flag
  if (bad_label && !valid && !(tg == pad_id || tg == eos_id || tg == seg0 + nseg)) bad_label[0] = 1;
  float m = -INFINITY, sum = 0.f, vl = 0.f, vall = 0.f;
  int pred = 0;
  for (int c = 0; c < nseg; ++c) {
    const float v = value(c);
    if (v > m) { sum = sum * __expf(m - v) + 1.f; m = v; pred = c; }
    else sum += __expf(v - m);
    if (c == label) vl = v;
    vall += v;
  }
  const float lse = m + __logf(sum);
  // F.cross_entropy(label_smoothing = eps): (1 - eps) * nll(label) + eps * mean_c nll(c)   (seg_criterion.py:269-287 with
  // --label-smoothing; eps == 0 keeps the plain expression, bit for bit)
  float lpix = valid ? (lse - vl) : 0.f, cnt = valid ? 1.f : 0.f;
  if (eps != 0.f && valid) lpix = (1.f - eps) * (lse - vl) + eps * (lse - vall / nseg);
  const float hot = 1.f - eps, unif = eps / nseg;
  if (valid) {
    atomicAdd(&sHist[nseg + pred], 1);
    atomicAdd(&sHist[2 * nseg + label], 1);
    if (pred == label) atomicAdd(&sHist[label], 1);
  }
  lpix = warp_sum(lpix); cnt = warp_sum(cnt);
  if ((tid & 63) == 0) { sRed[tid >> 6] = lpix; sRed[4 + (tid >> 6)] = cnt; }

  // ---- gradient: d value[c] = softmax - onehot (masked); adjoint of the interpolation
  float* tp = tile_partial + (long long)blockIdx.x * 9 * nseg;
  for (int c0 = 0; c0 < nseg; c0 += CH) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      const int c = c0 + cc;
      float d = 0.f;
      if (valid && c < nseg) d = __expf(value(c) - lse) - (c == label ? hot : 0.f) - unif;
      sD[tid][cc] = d;
    }
    __syncthreads();
    for (int i = tid; i < TS * 3 * CH; i += 256) {          // E[py][kx][cc] = sum_px WX[px][kx] D[py,px][cc]
      const int cc = i % CH, kx = (i / CH) % 3, yy = i / (3 * CH);
      float e = 0.f;
#pragma unroll
      for (int xx = 0; xx < TS; ++xx) e += sWX[xx][kx] * sD[yy * TS + xx][cc];
      sE[yy][kx][cc] = e;
    }
    __syncthreads();
    if (tid < 9 * CH) {                                      // G[ky][kx][cc] = sum_py WY[py][ky] E[py][kx][cc]
      const int cc = tid % CH, k = tid / CH, ky = k / 3, kx = k % 3;
      float gsum = 0.f;
#pragma unroll
      for (int yy = 0; yy < TS; ++yy) gsum += sWY[yy][ky] * sE[yy][kx][cc];
      if (c0 + cc < nseg) tp[k * nseg + c0 + cc] = gsum;
    }
    __syncthreads();
  }
  float* sp = stats_part + (long long)blockIdx.x * nstat;
  if (tid == 0) { sp[0] = sRed[0] + sRed[1] + sRed[2] + sRed[3]; sp[1] = sRed[4] + sRed[5] + sRed[6] + sRed[7]; }
  for (int i = tid; i < 3 * nseg; i += 256) sp[2 + i] = (float)sHist[i];
}

// dlogits[b, cell, c] = (1/Nvalid) * sum over the <=9 tiles whose 3x3 stencil contains `cell`
__global__ void seg_loss_gather_kernel(const float* tile_partial, const float* stats, bf16_t* dlogits, int ldl,
                                       long long dbs, int B, int hp, int wp, int nseg, float* loss_out) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int ncell = hp * wp;
  const long long total = (long long)B * (ncell + 1) * ldl;
  if (gid == 0) loss_out[0] = stats[1] > 0.f ? stats[0] / stats[1] : 0.f;
  if (gid >= total) return;
  const int c = (int)(gid % ldl);
  const long long r = gid / ldl;
  const int cell = (int)(r % (ncell + 1)), b = (int)(r / (ncell + 1));
  float g = 0.f;
  if (cell < ncell && c < nseg) {
    const int cy = cell / wp, cx = cell % wp;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = cy - (ky - 1);
      if (ty < 0 || ty >= hp) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = cx - (kx - 1);
        if (tx < 0 || tx >= wp) continue;
        g += tile_partial[(((long long)b * ncell + ty * wp + tx) * 9 + ky * 3 + kx) * nseg + c];
      }
    }
    g *= stats[1] > 0.f ? 1.f / stats[1] : 0.f;
  }
  dlogits[b * dbs + (long long)cell * ldl + c] = f2bf(g);
}

}  // namespace

extern "C" int ifseg_seg_loss_tiles(const void* logits, int ldl, long long logits_bs, const long long* target,
                                    long long target_bs, int B, int hp, int wp, int H, int W, int nseg,
                                    long long seg_id_offset, long long pad_id, long long eos_id,
                                    float* tile_partial, float* stats_part, int* bad_label, float label_smoothing,
                                    void* stream) {
  (void)hipGetLastError();
  if (H != hp * TS || W != wp * TS || nseg > NS_MAX || nseg < 1) return IFSEG_ERR_BAD_SHAPE;
  if (!(label_smoothing >= 0.f && label_smoothing <= 1.f)) return IFSEG_ERR_BAD_ARG;
  const size_t dyn = (size_t)(9 * (nseg | 1) + 3 * nseg) * 4;
  hipLaunchKernelGGL(seg_loss_tile_kernel, dim3(B * hp * wp), dim3(256), dyn, (hipStream_t)stream,
                     (const bf16_t*)logits, ldl, logits_bs, target, target_bs, hp, wp, W, nseg, seg_id_offset, pad_id,
                     eos_id, tile_partial, stats_part, bad_label, label_smoothing);
  IFSEG_CHECK_LAUNCH();
  return 0;
}

extern "C" int ifseg_seg_loss_gather(const float* tile_partial, const float* stats, void* dlogits, int ldl,
                                     long long dlogits_bs, int B, int hp, int wp, int nseg, float* loss_out,
                                     void* stream) {
  (void)hipGetLastError();
  const long long total = (long long)B * (hp * wp + 1) * ldl;
  hipLaunchKernelGGL(seg_loss_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     tile_partial, stats, (bf16_t*)dlogits, ldl, dlogits_bs, B, hp, wp, nseg, loss_out);
  IFSEG_CHECK_LAUNCH();
  return 0;
}
