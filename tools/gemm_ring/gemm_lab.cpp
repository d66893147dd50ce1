// GEMM laboratory: times and checks the tile configurations of the persistent ring kernel (csrc/gemm_ring.hip) against the
// one-tile-per-workgroup kernel (csrc/gemm.hip) through the C ABI, on the shapes the model launches.  No PyTorch: the binary
// starts at once on a GPU box.  Build: tools/build_lab.sh.   Run: tools/bin/gemm_lab [mode] [iters]
//   mode: perf (default) | check | ksweep | group
// Every configuration's output must be BIT-identical to configuration 0 (same MFMA sequence per output element).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../include/ifseg_hip.h"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16;
static bf16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16)(u >> 16); }
static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xffffff) / 8388608.f - 1.f; }

struct Buf {
  void* d = nullptr; size_t bytes = 0;
  void alloc(size_t b) { bytes = b; HC(hipMalloc(&d, b)); }
  void fill_bf16(size_t n, float scale) {
    std::vector<bf16> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(urand() * scale);
    HC(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  }
  void fill_f32(size_t n, float scale, float add = 0.f) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = urand() * scale + add;
    HC(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  }
  std::vector<unsigned char> host() const { std::vector<unsigned char> h(bytes); HC(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost)); return h; }
  void poison() { HC(hipMemset(d, 0x7f, bytes)); }
  ~Buf() { if (d) (void)hipFree(d); }
};

static void set_cfg(int cfg, bool group = false) {
  char b[16]; snprintf(b, sizeof b, "%d", cfg);
  setenv(group ? "IFSEG_GEMM_RING_GROUP" : "IFSEG_GEMM_RING", b, 1);
}

template <class F>
static double time_us(F&& fn, int iters) {
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  HC(hipDeviceSynchronize());
  // median of 5 batches
  std::vector<double> t;
  for (int r = 0; r < 5; ++r) {
    HC(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    HC(hipEventRecord(e1, 0));
    HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms * 1e3 / iters);
  }
  std::sort(t.begin(), t.end());
  HC(hipEventDestroy(e0)); HC(hipEventDestroy(e1));
  return t[2];
}

static std::vector<int> CFGS = {0, 6, 1, 3, 4, 5, 7};
static const char* cfg_name(int c) {
  switch (c) {
    case 0: return "tile 128x128 (gemm.hip)";
    case 1: return "ring 128x128x64 S3";
    case 3: return "ring 256x128x64 S2";
    case 4: return "ring 128x256x64 S2";
    case 5: return "ring 256x256x32 S4";
    case 6: return "ring 128x128x64 S2 2/CU";
    case 7: return "ring 256x256x64 S2";
  }
  return "?";
}

struct Case { int layout, M, N, K; int epi; };   // epi: 0 plain, 1 bias+alpha+resid, 2 relu+bias, 3 f32 accumulate, 4 bf16 accumulate

static int run_gemm(const Case& c, const Buf& A, const Buf& B, Buf& C, const Buf& bias, const Buf& resid) {
  const int lda = c.layout == IFSEG_GEMM_TN ? c.M : c.K;
  const int ldb = c.layout == IFSEG_GEMM_NT ? c.K : c.N;
  const void* bp = (c.epi == 1 || c.epi == 2) ? bias.d : nullptr;
  const void* rp = c.epi == 1 ? resid.d : nullptr;
  const float alpha = c.epi == 1 ? 0.125f : 1.f;
  const int an = c.epi == 1 ? (c.N / 3 / 8) * 8 : -1;
  const int flags = (c.epi == 2 ? IFSEG_GEMM_RELU : 0) | (c.epi == 3 ? (IFSEG_GEMM_OUT_F32 | IFSEG_GEMM_ACCUMULATE) : 0) |
                    (c.epi == 4 ? IFSEG_GEMM_ACCUMULATE : 0);
  return ifseg_gemm_bf16(c.layout, A.d, B.d, C.d, c.M, c.N, c.K, lda, ldb, c.N, bp, alpha, an, rp, c.N, flags, 1, 0, 0, 0, 0, 1, nullptr);
}

static void bench_cases(const std::vector<Case>& cases, int iters, bool check_only) {
  for (const Case& c : cases) {
    Buf A, B, C, bias, resid, C0;
    A.alloc((size_t)c.M * c.K * 2); A.fill_bf16((size_t)c.M * c.K, 1.f);
    B.alloc((size_t)c.N * c.K * 2); B.fill_bf16((size_t)c.N * c.K, 1.f);
    const size_t cel = (size_t)c.M * c.N, cbytes = cel * (c.epi == 3 ? 4 : 2);
    C.alloc(cbytes); C0.alloc(cbytes);
    if (c.epi == 3) C0.fill_f32(cel, 1.f); else C0.fill_bf16(cel, 1.f);
    bias.alloc(c.N * 2); bias.fill_bf16(c.N, 1.f);
    resid.alloc(cel * 2); resid.fill_bf16(cel, 1.f);
    std::vector<unsigned char> ref;
    const double gf = 2.0 * c.M * c.N * c.K * 1e-9;
    const char* ln = c.layout == 0 ? "NT" : c.layout == 1 ? "NN" : "TN";
    for (int cfg : CFGS) {
      set_cfg(cfg);
      HC(hipMemcpy(C.d, C0.d, cbytes, hipMemcpyDeviceToDevice));
      int rc = run_gemm(c, A, B, C, bias, resid);
      HC(hipDeviceSynchronize());
      if (rc) { printf("%s M%d N%d K%d epi%d  %-24s rc=%d\n", ln, c.M, c.N, c.K, c.epi, cfg_name(cfg), rc); continue; }
      std::vector<unsigned char> out = C.host();
      size_t bad = 0;
      if (cfg == 0) ref = out;
      else if (!ref.empty()) {
        const size_t el = c.epi == 3 ? 4 : 2;
        for (size_t i = 0; i < cel; ++i) bad += memcmp(&out[i * el], &ref[i * el], el) != 0;
      }
      double us = 0;
      if (!check_only && c.epi != 3 && c.epi != 4) us = time_us([&] { run_gemm(c, A, B, C, bias, resid); }, iters);
      printf("%s M%-5d N%-5d K%-5d epi%d  %-24s %8.1f us %7.1f TF/s  %s\n", ln, c.M, c.N, c.K, c.epi, cfg_name(cfg), us,
             us > 0 ? gf / us * 1e3 : 0.0, cfg == 0 ? "ref" : ref.empty() ? "-" : bad ? "MISMATCH" : "bit-equal");
      if (bad) printf("    mismatching elements: %zu of %zu\n", bad, cel);
      fflush(stdout);
    }
  }
}

static void bench_group(int iters) {
  // the weight gradients of one encoder layer: dW = dY^T X for qkv, out_proj, fc1, fc2 (tokens = 8480)
  const int T = 8480;
  struct P { int M, N; } ps[4] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};
  Buf dy[4], x[4], out[4];
  ifseg_gemm_tn_problem pr[4];
  double gf = 0;
  for (int i = 0; i < 4; ++i) {
    dy[i].alloc((size_t)T * ps[i].M * 2); dy[i].fill_bf16((size_t)T * ps[i].M, 1.f);
    x[i].alloc((size_t)T * ps[i].N * 2); x[i].fill_bf16((size_t)T * ps[i].N, 1.f);
    out[i].alloc(((size_t)ps[i].M * ps[i].N + ps[i].M) * 2);
    pr[i] = {dy[i].d, x[i].d, out[i].d, ps[i].M, ps[i].N, T, ps[i].M, ps[i].N, 1, 0};
    gf += 2.0 * ps[i].M * ps[i].N * T * 1e-9;
  }
  std::vector<std::vector<unsigned char>> ref(4);
  for (int cfg : {0, 1, 3, 4}) {
    set_cfg(cfg, true);
    for (int i = 0; i < 4; ++i) out[i].poison();
    int rc = ifseg_gemm_tn_group(4, pr, 256, nullptr);
    HC(hipDeviceSynchronize());
    size_t bad = 0;
    for (int i = 0; i < 4; ++i) {
      auto h = out[i].host();
      if (cfg == 0) ref[i] = h;
      else for (size_t e = 0; e < h.size() / 2; ++e) bad += memcmp(&h[e * 2], &ref[i][e * 2], 2) != 0;
    }
    double us = time_us([&] { ifseg_gemm_tn_group(4, pr, 256, nullptr); }, iters);
    printf("TN group (encoder layer, %.0f GF)  %-24s rc=%d %8.1f us %7.1f TF/s  %s\n", gf, cfg_name(cfg), rc, us, gf / us * 1e3,
           cfg == 0 ? "ref" : bad ? "MISMATCH" : "bit-equal");
    if (bad) printf("    mismatching elements: %zu\n", bad);
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "perf";
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  if (argc > 3) {          // explicit configuration list: "0,6,3"
    CFGS.clear();
    for (char* t = strtok(argv[3], ","); t; t = strtok(nullptr, ",")) CFGS.push_back(atoi(t));
  }
  HC(hipSetDevice(0));
  printf("abi %d\n", ifseg_abi_version());
  if (mode == "check") {
    // ragged shapes, every epilogue; outputs only compared
    std::vector<Case> cs;
    for (int epi = 0; epi <= 4; ++epi) {
      cs.push_back({IFSEG_GEMM_NT, 1000, 776, 136, epi});
      cs.push_back({IFSEG_GEMM_NN, 1000, 776, 136, epi});
      cs.push_back({IFSEG_GEMM_NT, 8480, 768, 768, epi});
    }
    cs.push_back({IFSEG_GEMM_NT, 130, 2304, 768, 1});
    cs.push_back({IFSEG_GEMM_NN, 2050, 1536, 72, 0});
    cs.push_back({IFSEG_GEMM_NT, 300, 136, 64, 2});
    bench_cases(cs, iters, true);
  } else if (mode == "perf") {
    std::vector<Case> cs;
    const int M = 8480;
    for (auto nk : {std::pair<int, int>{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}, {1536, 768}}) {
      cs.push_back({IFSEG_GEMM_NT, M, nk.first, nk.second, 1});
      cs.push_back({IFSEG_GEMM_NN, M, nk.first, nk.second, 0});
    }
    cs.push_back({IFSEG_GEMM_NN, M, 768, 2304, 0});
    cs.push_back({IFSEG_GEMM_NT, 4096, 4096, 4096, 0});
    bench_cases(cs, iters, false);
  } else if (mode == "ksweep") {
    std::vector<Case> cs;
    for (int N : {768, 3072})
      for (int K : {64, 256, 768, 1536, 3072}) cs.push_back({IFSEG_GEMM_NT, 8480, N, K, 1});
    bench_cases(cs, iters, false);
  } else if (mode == "group") {
    bench_group(iters);
  } else if (mode == "ablate") {
    // ablate <iters> <cfgs>: the k-loop with parts left out (wrong results; what each part costs)
    const char* names[] = {"full", "no MFMA", "no DMA", "no MFMA, no DMA", "no LDS reads", "no MFMA, no LDS reads (DMA only)", "no DMA, no LDS reads (MFMA only)", "barriers only"};
    for (auto sh : {std::pair<int, int>{768, 3072}, {3072, 768}}) {
      for (int ab = 0; ab < 8; ++ab) {
        char b[8]; snprintf(b, sizeof b, "%d", ab); setenv("IFSEG_RING_ABLATE", b, 1);
        printf("---- ablation %d: %s\n", ab, names[ab]);
        bench_cases({{IFSEG_GEMM_NT, 8480, sh.first, sh.second, 0}}, iters, false);
      }
    }
    unsetenv("IFSEG_RING_ABLATE");
  } else if (mode == "one") {
    // one <iters> <cfgs> <layout> <M> <N> <K> [epi]: a single shape (profiling)
    if (argc < 8) { fprintf(stderr, "one <iters> <cfgs> <layout> <M> <N> <K> [epi]\n"); return 2; }
    bench_cases({{atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), argc > 8 ? atoi(argv[8]) : 0}}, iters, false);
  }
  return 0;
}
