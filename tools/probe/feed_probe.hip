// Per-CU operand feed: how many bytes per clock one workgroup (4 waves) can pull from L2 / HBM into (a) LDS by LDS-DMA,
// (b) VGPRs by buffer_load_dwordx4, (c) VGPRs and on into LDS by ds_write_b128.  Access pattern of a GEMM A tile: 8 rows x 128 B per
// wave instruction, row stride `ld` bytes, 32 KiB per k-step and workgroup.  hipcc --offload-arch=gfx950 -O3 feed_probe.hip -o feed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i32 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  v4i32 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA ring of S stages; 1: buffer_load to VGPR (S = stages of 8 x 16 B per lane in flight); 2: VGPR + ds_write_b128
template <int MODE, int S>
__global__ __launch_bounds__(256, 2) void feed_kernel(const unsigned char* src, unsigned bytes, int ld, int nk, int rows_per_wg, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[S * 32768];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const v4i32 rs = make_rsrc(src, bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // piece p (0..31) of a k-step: rows 8p..8p+7 of the workgroup's 256 rows (two 128-row operand tiles), 128 B each
  unsigned off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = blockIdx.x * rows_per_wg + (wave * 8 + i) * 8 + (lane >> 3);
    off[i] = (unsigned)row * (unsigned)ld + (lane & 7) * 16;
  }
  unsigned acc = 0;
  if constexpr (MODE == 0) {
    auto issue = [&](int kt, int st) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned d = lds0 + st * 32768 + (wave * 8 + i) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(d), "v"(off[i]), "s"(rs), "s"(kt * 128) : "memory");
      }
    };
    for (int p = 0; p < S - 1 && p < nk; ++p) issue(p, p);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + S - 1 < nk) vm_wait<(S - 2) * 8>(); else vm_wait<0>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt + S - 1 < nk) issue(kt + S - 1, (st + S - 1) % S);
      acc += *reinterpret_cast<const unsigned*>(smem + st * 32768 + threadIdx.x * 4);
      st = (st + 1) % S;
    }
  } else {
    v4i32 r[S][8];
#define ISSUE(kt, st)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                   \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[st][i]) : "v"(off[i]), "s"(rs), "s"((kt) * 128) : "memory");
    static_assert(S == 2, "two register stages");
    ISSUE(0, 0)
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 1 < nk) { ISSUE(kt + 1, 1) }
      if (kt + 1 < nk) vm_wait<8>(); else vm_wait<0>();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("" : "+v"(r[0][i]));
        if (MODE == 2) *reinterpret_cast<v4i32*>(smem + (wave * 8 + i) * 1024 + lane * 16) = r[0][i];
        else acc ^= r[0][i].x ^ r[0][i].w;
      }
      if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); acc += *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 4); }
      if (kt + 2 < nk) { ISSUE(kt + 2, 0) }
      if (kt + 1 < nk) {
        if (kt + 2 < nk) vm_wait<8>(); else vm_wait<0>();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("" : "+v"(r[1][i]));
          if (MODE == 2) *reinterpret_cast<v4i32*>(smem + 32768 + (wave * 8 + i) * 1024 + lane * 16) = r[1][i];
          else acc ^= r[1][i].x ^ r[1][i].w;
        }
        if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); acc += *reinterpret_cast<const unsigned*>(smem + 32768 + threadIdx.x * 4); }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int nk = 48, ld = 6144;                       // K = 3072 bf16
  const int rows_per_wg = 256;
  for (int wgs : {1, 8, 32, 64, 128, 256, 512, 1024}) {
    const size_t bytes = (size_t)wgs * rows_per_wg * ld;
    if (bytes >= (1ull << 31)) continue;
    unsigned char* src; unsigned* sink;
    HC(hipMalloc(&src, bytes)); HC(hipMemset(src, 1, bytes)); HC(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, (unsigned)bytes, ld, nk, rows_per_wg, sink);
      HC(hipDeviceSynchronize());
      HC(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, (unsigned)bytes, ld, nk, rows_per_wg, sink);
      HC(hipEventRecord(e1, 0)); HC(hipEventSynchronize(e1));
      float ms; HC(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / 20, per_wg = (double)nk * 32768;
      printf("wgs %4d  %-34s %8.1f us   %6.1f GB/s per workgroup   %7.2f TB/s total\n", wgs, name, us, per_wg / us * 1e-3, per_wg * wgs / us * 1e-6);
    };
    run("LDS-DMA, 2 stages", feed_kernel<0, 2>);
    run("LDS-DMA, 3 stages", feed_kernel<0, 3>);
    run("LDS-DMA, 4 stages", feed_kernel<0, 4>);
    run("buffer_load -> VGPR, 2 stages", feed_kernel<1, 2>);
    run("buffer_load -> VGPR -> ds_write", feed_kernel<2, 2>);
    HC(hipFree(src)); HC(hipFree(sink));
  }
  return 0;
}
