// register-corruption probe: every lane parks NV known values in VGPRs, idles, re-checks them and logs what changed
#include <hip/hip_runtime.h>
constexpr int NV = 48;
__global__ __launch_bounds__(256) void victim_kernel(unsigned* log, unsigned* count, int spins) {
  unsigned x[NV];
  const unsigned tag = (blockIdx.x * 256u + threadIdx.x) * 64u;
#pragma unroll
  for (int i = 0; i < NV; ++i) { x[i] = 0x40000000u + tag + i; asm volatile("" : "+v"(x[i])); }
  for (int s = 0; s < spins; ++s) {
    __builtin_amdgcn_s_sleep(8);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(x[i]));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (x[i] != 0x40000000u + tag + i) {
      const unsigned k = atomicAdd(count, 1u);
      if (k < 4096) { log[4 * k] = i; log[4 * k + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * k + 2] = x[i]; log[4 * k + 3] = 0x40000000u + tag + i; }
    }
  }
}
extern "C" int victim_launch(unsigned* log, unsigned* count, int blocks, int spins, void* stream) {
  hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, count, spins);
  return (int)hipGetLastError();
}

// load-return probe: src[i] == base + i ; every lane re-reads float4 chunks and checks them
__global__ __launch_bounds__(256) void victim_load_kernel(const unsigned* __restrict__ src, const unsigned* __restrict__ src2, int n, unsigned base, unsigned* log, unsigned* count, int reps) {
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < reps; ++r) {
    for (int k = lane * 8; k < n; k += 512) {
      const uint4 a = *reinterpret_cast<const uint4*>(src + k), b = *reinterpret_cast<const uint4*>(src + k + 4);
      const uint4 c = *reinterpret_cast<const uint4*>(src2 + k), d = *reinterpret_cast<const uint4*>(src2 + k + 4);
      const unsigned got[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned want = (e < 8 ? base : base + 0x100000u) + k + (e & 7);
        if (got[e] != want) {
          const unsigned q = atomicAdd(count, 1u);
          if (q < 4096) { log[4 * q] = e; log[4 * q + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * q + 2] = got[e]; log[4 * q + 3] = want; }
        }
      }
    }
  }
}
extern "C" int victim_load_launch(const unsigned* src, const unsigned* src2, int n, unsigned base, unsigned* log, unsigned* count, int blocks, int reps, void* stream) {
  hipLaunchKernelGGL(victim_load_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src2, n, base, log, count, reps);
  return (int)hipGetLastError();
}

// packed-fp32 ALU probe: the same recurrence through v_pk_fma_f32 and through two v_fma_f32; no memory traffic inside the loop
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void victim_pk_kernel(unsigned* log, unsigned* count, int iters, float seed) {
  const float t = (float)(threadIdx.x + 1) * 1e-3f + seed;
  f2 acc = {t, -t}, mul = {0.999f, 1.001f}, add = {t * 0.5f, t * 0.25f};
  float a0 = t, a1 = -t;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(mul), "v"(add));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a0), "v"(mul.x), "v"(add.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(a1), "v"(mul.y), "v"(add.y));
  }
  const bool b0 = __float_as_uint(acc.x) != __float_as_uint(a0), b1 = __float_as_uint(acc.y) != __float_as_uint(a1);
  if (b0 || b1) {
    const unsigned q = atomicAdd(count, 1u);
    if (q < 4096) { log[4 * q] = (b0 ? 1 : 0) | (b1 ? 2 : 0); log[4 * q + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * q + 2] = __float_as_uint(b0 ? acc.x : acc.y); log[4 * q + 3] = __float_as_uint(b0 ? a0 : a1); }
  }
}
extern "C" int victim_pk_launch(unsigned* log, unsigned* count, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(victim_pk_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, count, iters, 0.5f);
  return (int)hipGetLastError();
}

// load-ORDER probe: four loads, each consumed right after "s_waitcnt vmcnt(3 - i)"; a register still holding the sentinel (or a
// previous round's value) means a younger load was counted as returned before an older one
__global__ __launch_bounds__(256) void victim_order_kernel(const unsigned* __restrict__ p0, const unsigned* __restrict__ p1, const unsigned* __restrict__ p2,
                                                           const unsigned* __restrict__ p3, int n, unsigned* log, unsigned* count, int reps) {
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < reps; ++r) {
    for (int k = lane; k < n; k += 64) {
      unsigned r0 = 0xdead0000u, r1 = 0xdead0001u, r2 = 0xdead0002u, r3 = 0xdead0003u, t0, t1, t2, t3;
      const unsigned *a0 = p0 + k, *a1 = p1 + k, *a2 = p2 + k, *a3 = p3 + k;
      asm volatile(
          "global_load_dword %0, %8, off\n\t"
          "global_load_dword %1, %9, off\n\t"
          "global_load_dword %2, %10, off\n\t"
          "global_load_dword %3, %11, off\n\t"
          "s_waitcnt vmcnt(3)\n\tv_mov_b32 %4, %0\n\t"
          "s_waitcnt vmcnt(2)\n\tv_mov_b32 %5, %1\n\t"
          "s_waitcnt vmcnt(1)\n\tv_mov_b32 %6, %2\n\t"
          "s_waitcnt vmcnt(0)\n\tv_mov_b32 %7, %3\n\t"
          : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
      const unsigned got[4] = {t0, t1, t2, t3};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned want = 0x40000000u + e * 0x100000u + k;
        if (got[e] != want) {
          const unsigned q = atomicAdd(count, 1u);
          if (q < 4096) { log[4 * q] = e; log[4 * q + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * q + 2] = got[e]; log[4 * q + 3] = want; }
        }
      }
    }
  }
}
extern "C" int victim_order_launch(const unsigned* p0, const unsigned* p1, const unsigned* p2, const unsigned* p3, int n, unsigned* log, unsigned* count,
                                   int blocks, int reps, void* stream) {
  hipLaunchKernelGGL(victim_order_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p0, p1, p2, p3, n, log, count, reps);
  return (int)hipGetLastError();
}

// the load schedule of ffn_ln_coef_kernel: five 16-byte loads in flight, consumed behind vmcnt(4) / (2) / (1) / (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void victim_sched_kernel(const unsigned* __restrict__ pw, const unsigned* __restrict__ pg, const unsigned* __restrict__ pb,
                                                           int J, unsigned* log, unsigned* count) {
  __shared__ __attribute__((aligned(16))) unsigned park[256 * 20];
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= J) return;
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(park + threadIdx.x * 20);
  for (int it = 0; it < 6; ++it) {
    const unsigned* aw = pw + (long long)j * 1536 + it * 256 + lane * 4;      // 16 B per lane
    const unsigned* ag = pg + it * 512 + lane * 8;                            // 32 B per lane
    const unsigned* ab = pb + it * 512 + lane * 8;
    u4 r0 = {0xdead0000u, 0xdead0000u, 0xdead0000u, 0xdead0000u}, r1 = r0, r2 = r0, r3 = r0, r4 = r0;
    asm volatile(
        "global_load_dwordx4 %0, %5, off\n\t"
        "global_load_dwordx4 %1, %7, off\n\t"
        "global_load_dwordx4 %2, %6, off\n\t"
        "global_load_dwordx4 %3, %6, off offset:16\n\t"
        "global_load_dwordx4 %4, %7, off offset:16\n\t"
        "s_waitcnt vmcnt(4)\n\tds_write_b128 %8, %0\n\t"
        "s_waitcnt vmcnt(2)\n\tds_write_b128 %8, %1 offset:16\n\tds_write_b128 %8, %2 offset:32\n\t"
        "s_waitcnt vmcnt(1)\n\tds_write_b128 %8, %3 offset:48\n\t"
        "s_waitcnt vmcnt(0)\n\tds_write_b128 %8, %4 offset:64\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)
        : "v"(aw), "v"(ag), "v"(ab), "v"(lds) : "memory");
    const unsigned* mine = park + threadIdx.x * 20;
    const unsigned wbase = 0x10000000u + j * 1536 + it * 256 + lane * 4, gbase = 0x40000000u + it * 512 + lane * 8, bbase = 0x50000000u + it * 512 + lane * 8;
    const unsigned want0[5] = {wbase, bbase, gbase, gbase + 4, bbase + 4};
#pragma unroll
    for (int q5 = 0; q5 < 5; ++q5)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned got = mine[q5 * 4 + e], want = want0[q5] + e;
        if (got != want) {
          const unsigned q = atomicAdd(count, 1u);
          if (q < 4096) { log[4 * q] = q5 * 4 + e; log[4 * q + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * q + 2] = got; log[4 * q + 3] = want; }
        }
      }
  }
}
extern "C" int victim_sched_launch(const unsigned* pw, const unsigned* pg, const unsigned* pb, int J, unsigned* log, unsigned* count, int layers, void* stream) {
  hipLaunchKernelGGL(victim_sched_kernel, dim3((J + 3) / 4, layers), dim3(256), 0, (hipStream_t)stream, pw, pg, pb, J, log, count);
  return (int)hipGetLastError();
}

// packed fp32 with operand-half selection (what the SLP-vectorised coef loop uses)
__global__ __launch_bounds__(256) void victim_pksel_kernel(unsigned* log, unsigned* count, int iters, float seed) {
  const float t = (float)(threadIdx.x + 1) * 1e-3f + seed;
  f2 acc = {t, -t}, mul = {0.999f, 1.001f}, add = {t * 0.5f, t * 0.25f}, tmp, mv;
  float a0 = t, a1 = -t, m0, m1;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(mv) : "v"(add), "v"(mul));                   // (add.y, mul.x)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tmp) : "v"(mv), "v"(acc));   // (add.y*acc.y, mul.x*acc.x)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(acc) : "v"(acc), "v"(mul), "v"(tmp));   // (acc.x*mul.x+tmp.x, acc.y*mul.x+tmp.y)
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(add.y), "v"(a1));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(mul.x), "v"(a0));
    float n0, n1;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n0) : "v"(a0), "v"(mul.x), "v"(m0));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n1) : "v"(a1), "v"(mul.x), "v"(m1));
    a0 = n0 * 0.5f; a1 = n1 * 0.5f; acc.x *= 0.5f; acc.y *= 0.5f;
  }
  const bool b0 = __float_as_uint(acc.x) != __float_as_uint(a0), b1 = __float_as_uint(acc.y) != __float_as_uint(a1);
  if (b0 || b1) {
    const unsigned q = atomicAdd(count, 1u);
    if (q < 4096) { log[4 * q] = (b0 ? 1 : 0) | (b1 ? 2 : 0); log[4 * q + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * q + 2] = __float_as_uint(b0 ? acc.x : acc.y); log[4 * q + 3] = __float_as_uint(b0 ? a0 : a1); }
  }
}
extern "C" int victim_pksel_launch(unsigned* log, unsigned* count, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(victim_pksel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, count, iters, 0.5f);
  return (int)hipGetLastError();
}

// does the UNSELECTED half of a packed-fp32 source influence the result?  src1 = {x, pattern}, op_sel_hi:[1,0,1] (both lanes use x)
__global__ void pk_unused_half_kernel(const unsigned* patterns, int np, unsigned* out) {
  const float t = (float)(threadIdx.x + 1) * 1e-3f + 0.5f;
  for (int p = 0; p < np; ++p) {
    f2 a = {t, -t}, c = {t * 0.5f, t * 0.25f}, b, d;
    b.x = 0.999f; b.y = __uint_as_float(patterns[p]);
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    float r0, r1;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a.x), "v"(b.x), "v"(c.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a.y), "v"(b.x), "v"(c.y));
    if (__float_as_uint(d.x) != __float_as_uint(r0)) atomicAdd(out + 2 * p, 1u);
    if (__float_as_uint(d.y) != __float_as_uint(r1)) atomicAdd(out + 2 * p + 1, 1u);
  }
}
extern "C" int pk_unused_half_launch(const unsigned* patterns, int np, unsigned* out, void* stream) {
  hipLaunchKernelGGL(pk_unused_half_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, patterns, np, out);
  return (int)hipGetLastError();
}

// v0..v3 parked by hand (one asm block: the compiler cannot touch them in between)
__global__ __launch_bounds__(256) void victim_v0_kernel(unsigned* log, unsigned* count, int spins) {
  unsigned r0, r1, r2, r3;
  const unsigned init = 0x40000000u;
  asm volatile(
      "v_mov_b32 v0, %4\n\tv_mov_b32 v1, %4\n\tv_mov_b32 v2, %4\n\tv_mov_b32 v3, %4\n\t"
      "s_mov_b32 s20, %5\n"
      "1:\n\ts_sleep 2\n\t"
      "v_add_u32 v0, 1, v0\n\tv_add_u32 v1, 1, v1\n\tv_add_u32 v2, 1, v2\n\tv_add_u32 v3, 1, v3\n\t"
      "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
      "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3"
      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(init), "s"(spins) : "v0", "v1", "v2", "v3", "s20", "scc");
  const unsigned got[4] = {r0, r1, r2, r3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (got[i] != init + spins) {
      const unsigned k = atomicAdd(count, 1u);
      if (k < 4096) { log[4 * k] = i; log[4 * k + 1] = blockIdx.x * 256u + threadIdx.x; log[4 * k + 2] = got[i]; log[4 * k + 3] = init + spins; }
    }
  }
}
extern "C" int victim_v0_launch(unsigned* log, unsigned* count, int blocks, int spins, void* stream) {
  hipLaunchKernelGGL(victim_v0_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, count, spins);
  return (int)hipGetLastError();
}
