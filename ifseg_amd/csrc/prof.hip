#include "prof.h"
#include "../../include/ifseg_hip.h"
#include <vector>

namespace {
struct Slot {
  std::vector<hipEvent_t> ev;   // pairs
  double flops = 0, bytes = 0;
  int launches = 0;
};
Slot g_slot[IFSEG_K_COUNT];
unsigned g_mask = 0;
int g_stride = 1;                 // time every g_stride-th launch of a family (an event pair costs two queue packets)
int g_seen[IFSEG_K_COUNT];
bool g_open[IFSEG_K_COUNT];
}  // namespace

void ifseg_prof_begin(int kind, hipStream_t s, double flops, double bytes) {
  if (!(g_mask & (1u << kind))) return;
  g_open[kind] = (g_seen[kind]++ % g_stride) == 0;
  if (!g_open[kind]) return;
  Slot& sl = g_slot[kind];
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  sl.ev.push_back(a); sl.ev.push_back(b);
  sl.flops += flops; sl.bytes += bytes; sl.launches++;
  (void)hipEventRecord(a, s);
}
void ifseg_prof_end(int kind, hipStream_t s) {
  if (!(g_mask & (1u << kind)) || !g_open[kind]) return;
  g_open[kind] = false;
  Slot& sl = g_slot[kind];
  if (sl.ev.size() >= 2) (void)hipEventRecord(sl.ev.back(), s);
}

extern "C" int ifseg_prof_enable(unsigned mask) { g_mask = mask; return 0; }
extern "C" int ifseg_prof_stride(int stride) { g_stride = stride > 0 ? stride : 1; return 0; }

extern "C" int ifseg_prof_reset(void) {
  (void)hipGetLastError();
  for (auto& sl : g_slot) {
    for (auto e : sl.ev) (void)hipEventDestroy(e);
    sl.ev.clear(); sl.flops = sl.bytes = 0; sl.launches = 0;
  }
  for (int k = 0; k < IFSEG_K_COUNT; ++k) { g_seen[k] = 0; g_open[k] = false; }
  return 0;
}

// total elapsed ms over all recorded launches of `kind` (synchronises on the events)
extern "C" int ifseg_prof_read(int kind, double* ms, double* flops, double* bytes, int* launches) {
  (void)hipGetLastError();
  if (kind < 0 || kind >= IFSEG_K_COUNT) return IFSEG_ERR_BAD_ARG;
  Slot& sl = g_slot[kind];
  double tot = 0;
  for (size_t i = 0; i + 1 < sl.ev.size(); i += 2) {
    if (hipEventSynchronize(sl.ev[i + 1]) != hipSuccess) return IFSEG_ERR_BAD_ARG;
    float t = 0;
    if (hipEventElapsedTime(&t, sl.ev[i], sl.ev[i + 1]) == hipSuccess) tot += t;
  }
  *ms = tot; *flops = sl.flops; *bytes = sl.bytes; *launches = sl.launches;
  return 0;
}

