// Optional per-kernel-family timing with HIP events on the launch stream (used by
// bench.py for the roofline object).  Disabled by default: zero overhead.
#pragma once
#include <hip/hip_runtime.h>

enum {
  IFSEG_K_GEMM_NT = 0, IFSEG_K_GEMM_NN, IFSEG_K_GEMM_TN, IFSEG_K_CONV, IFSEG_K_ATTN_FWD, IFSEG_K_ATTN_DKV,
  IFSEG_K_ATTN_DQ, IFSEG_K_LN_FWD, IFSEG_K_LN_BWD, IFSEG_K_STEM, IFSEG_K_ADAM, IFSEG_K_COUNT
};

void ifseg_prof_begin(int kind, hipStream_t s, double flops, double bytes);
void ifseg_prof_end(int kind, hipStream_t s);
