/* ifseg_hip.h -- C ABI of libifseg_hip.so (MI355X / gfx950).
 *
 * The reference (alinlab/ifseg) has no FFI on this path: its hot path is
 * PyTorch Python calling stock ATen ops.  This library is the layer that sits
 * *below* the reference's nn.Module boundary (models/segofa/ (all .py files)): every entry
 * point replaces a group of ATen calls the reference makes, cited per function
 * as file:line under /root/reference.  Plain pointers and sizes only; all
 * pointers are DEVICE pointers unless noted; `stream` is a hipStream_t.
 * Every function returns 0 on success, a hipError_t value (>0) if the launch
 * failed, or one of the IFSEG_ERR_* codes (<0) for argument errors.  Nothing
 * here allocates, synchronises or falls back to the host.
 *
 * Layout conventions: activations are bf16, token-major [rows, C] with
 * row = b*T + t ("batch-first"); heads are column blocks of 64 (h*64..h*64+63).
 */
#ifndef IFSEG_HIP_H
#define IFSEG_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define IFSEG_ABI_VERSION 18
#define IFSEG_ERR_BAD_SHAPE (-2)
#define IFSEG_ERR_BAD_ARG (-3)

int ifseg_abi_version(void);
/* non-zero when the library was compiled with a measurement switch that leaves work out or changes results (IFSEG_EXP_*,
 * RING_ABLATE / RING_NOBAR / RING_NOWAIT builds of tools/variant.py): bench.py refuses to report a number from such a build */
int ifseg_experimental_build(void);

/* ------------------------------------------------------------------ GEMM */
#define IFSEG_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T   (F.linear forward)        */
#define IFSEG_GEMM_NN 1 /* C[M,N] = A[M,K] . B[K,N]     (dX = dY . W)             */
#define IFSEG_GEMM_TN 2 /* C[M,N] = A[K,M]^T . B[K,N]   (dW = dY^T . X)           */
#define IFSEG_GEMM_RELU 1
#define IFSEG_GEMM_OUT_F32 2
#define IFSEG_GEMM_ACCUMULATE 4 /* C += result */
#define IFSEG_GEMM_COLSUM 8     /* TN, ldc == N: also sum_k A[k][m] -- the bias gradient of a Linear whose weight gradient
                                   is this GEMM (dW = dY^T X, db = colsum dY).  splitk > 1: every k-slice slab is
                                   [M x N | round4(M)] floats (summed by the same ifseg_reduce_parts pass); splitk == 1 (bf16
                                   output): db is written as M bf16 right behind dW [M x N] (honours ACCUMULATE) */

/* bf16 MFMA GEMM, fp32 accumulate, epilogue
 *   C = ((A.B + bias[n]) * alpha[for n < alpha_ncols]) + resid[m,n]  (+relu)
 * Replaces nn.Linear / F.linear and their autograd (addmm / mm) as used by
 * q/k/v/out_proj (unify_multihead_attention.py:327-346,513), fc1/fc2
 * (unify_transformer_layer.py:279-283,556-560), image_proj
 * (encoder_module.py:416), pos_{q,k}_linear (encoder_module.py:765-770,
 * decoder_module.py:350-363); `alpha` carries the reference's `q *= scaling`
 * (unify_multihead_attention.py:346) and `* pos_scaling`; `resid` the
 * residual_connection (unify_transformer_layer.py:196,289).
 * N, lda, ldb must be multiples of 8 (16-byte rows); bias/resid bf16.
 * splitk > 1: the reduction is cut into ceil(K/kchunk) slices (kchunk = K/splitk rounded up
 * to 64); slice z writes its partial product to the fp32 workspace C + z*M*ldc (requires
 * IFSEG_GEMM_OUT_F32, no epilogue terms); the caller sums the slices (ifseg_reduce_parts).
 * Used for the weight-gradient GEMMs whose K is the token count. */
int ifseg_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                    int ldb, int ldc, const void* bias, float alpha, int alpha_ncols,
                    const void* resid, int ldr, int flags, int batch, long long strideA,
                    long long strideB, long long strideC, long long strideR, int splitk, void* stream);
/* Grouped weight-gradient GEMM: n <= IFSEG_GEMM_GROUP_MAX independent products C_i[M,N] = A_i[K,M]^T . B_i[K,N]
 * (dW = dY^T X of the Linear modules of ONE transformer layer: fc1, fc2, q|k|v, out_proj, cross q, cross k|v --
 * unify_transformer_layer.py:279-283,556-560, unify_multihead_attention.py:327-346,513 under autograd) in one launch,
 * one workgroup per 128x128 tile over all K tokens: no split-K slabs, no reduction launches.  C_i is bf16 [M, N]
 * (ldc == N); colsum != 0 also writes db_i[M] = column sums of A_i (bf16) right behind C_i (a Linear's weight and
 * bias are adjacent in the gradient arena); accumulate != 0 adds to what C_i / db_i hold. */
#define IFSEG_GEMM_GROUP_MAX 8
typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K, lda, ldb, colsum, accumulate;
} ifseg_gemm_tn_problem;
int ifseg_gemm_tn_group(int n, const ifseg_gemm_tn_problem* probs, int max_workgroups, void* stream);
/* max_workgroups > 0 caps the grid (a workgroup then walks several tiles): 256 = one per CU, so that a launch on a side
 * stream leaves half of every CU to the kernels of the dependent chain it runs next to; <= 0: one workgroup per tile. */
/* C[M,N] = A[M,K] . B[K,N] (NN, bf16 out) and, from the same epilogue, the per-head row dots with a second operand:
 *   dot_out[(m / rows_per_batch) * (N/64) + h][m % rows_per_batch] = sum_{c<64} C[m][64h+c] (as stored) * dot[m][64h+c]
 * i.e. the attention backward's delta = rowsum(dO * O) ([B,H,T] fp32) while dO = d(attn_ln input) . W_out is produced
 * (unify_multihead_attention.py:503-513 backward); N % 64 == 0.  Replaces phase 1 of ifseg_attn_bwd. */
int ifseg_gemm_nn_rowdot(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         const void* dot, int ldd, float* dot_out, int rows_per_batch, void* stream);

/* The FFN's ffn_layernorm(gelu(fc1(x))) -> fc2 on the way back (unify_transformer_layer.py:279-283 under autograd) without a
 * 3072-wide LayerNorm-backward pass.  dY [M, J] = gradient of the fc2 output t [M, J] (J = embed dim), W2 [J, N] (N = ffn
 * dim), u [M, N] the saved fc1 output, gamma / beta fp32 [N] of ffn_layernorm, mean / rstd [M] its saved row statistics:
 *   ifseg_ffn_ln_coef       coef[0][j] = sum_k gamma_k W2[j,k],  coef[1][j] = b2_j + sum_k beta_k W2[j,k]        (per step)
 *   ifseg_ffn_ln_rowstats   c[m][0] = mean_k(gamma_k dz_k), c[m][1] = mean_k(gamma_k xh_k dz_k) with dz = dY W2, computed as
 *                           row dots of dY with coef[0] and with (t - coef[1]) -- linear in dz, no 3072-wide tensor is read
 *   ifseg_gemm_nn_gelu_ln_bwd   C = du [M, N] (bf16) = rstd (gamma dz - c1 - xh c2) gelu'(u), dz = A . B in the accumulators
 *                           (A = dY [M, K = J], B = W2 [K, N]); dz never reaches HBM
 *   ifseg_ffn_ln_param_grads    dbeta_k = sum_j db2_j W2[j,k], dgamma_k = (sum_j W2[j,k] dW2[j,k] - beta_k dbeta_k) / gamma_k
 *                           from the (bf16) weight / bias gradient of fc2, written as bf16 */
int ifseg_ffn_ln_coef(const void* const* w2, int ldw, const float* const* gamma, const float* const* beta,
                      const void* const* b2 /* bf16 [J] each, or NULL */, float* const* coef /* [2][J] each */,
                      int L /* layers in this launch (host arrays of L device pointers), <= 32 */, int J, int N, void* stream);
int ifseg_ffn_ln_rowstats(const void* dy, int lddy, const void* t, int ldt, const float* coef, float* c /* [rows][2] */,
                          int rows, int J, int N, void* stream);
int ifseg_gemm_nn_gelu_ln_bwd(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                              const void* u, int ldu, const float* gamma, const float* mean, const float* rstd,
                              const float* cstats, void* stream);
int ifseg_ffn_ln_param_grads(const void* w2, const void* dw2, const void* db2, const float* gamma, const float* beta,
                             void* dgamma, void* dbeta, float* workspace /* >= 16 N floats: partial column sums of 8 row slabs */,
                             int J, int N,
                             /* rescue of the columns whose gain is too small to divide by (|gamma_k| < 0.05 max(1, |beta_k|), 0 included):
                              * dgamma_k = sum_m (dY W2)[m][k] xhat[m][k] from its definition.  dy [M, J] (the gradient of fc2's output), u [M, N]
                              * (fc1's output), mean / rstd [M] of ffn_layernorm; dy == NULL: no rescue (a zero gain then yields 0, not NaN) */
                             const void* dy, int lddy, const void* u, int ldu, const float* mean, const float* rstd, int M,
                             void* stream);

/* Implicit-GEMM conv on NHWC bf16 with folded FrozenBatchNorm (+residual, +ReLU).
 * `w` is [Cout][KH][KW][Cin] with the BN scale already folded in, `shift` the
 * folded BN bias.  Replaces Conv2d + FrozenBatchNorm2d + ReLU (+ identity add)
 * of Bottleneck.forward (resnet.py:117-137; frozen_bn.py:36-57). Cin % 64 == 0. */
int ifseg_conv2d_nhwc_bf16(const void* in, const void* w, const void* shift, const void* resid,
                           void* out, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int relu, void* stream);

/* ------------------------------------------------------------- attention */
/* Fused position-biased attention forward (flash-style, nothing [T,S]-sized in HBM):
 *   S = q k^T + pos_q pos_k^T + rel(i,j) (+causal mask);  O = softmax_fp32(S) v
 * q is expected pre-scaled by (2*64)^-0.5 and pos_q by pos_scaling (GEMM epilogue).
 * Replaces bmm + bias add + mask add + softmax + bmm of
 * unify_multihead_attention.py:459-501 together with the bias construction of
 * encoder_module.py:757-771,790-809 / decoder_module.py:335-366,553-558,601-631.
 * q,k,v,out: bf16, row stride ld*, batch stride *_bs (elements), head h at columns
 * [h*64, h*64+64).  pos_q [T,ldpq], pos_k [S,ldpk] batch-invariant (may be NULL).
 * lse: fp32 [B,H,T], the log-sum-exp of the scores in log2 units (natural lse x log2 e; the backward
 * consumes it with exp2).  Token order: grid tokens [0,P) then tail tokens [P,T)
 * (decoder: bos is moved to the end by the caller).  rel_mode=1:
 *   i,j <  P : rel2d[h][gcode[i]-gcode[j]+code_bias]     (image / seg grid)
 *   i,j >= P : rel1d[h][(i-j)+Lt-1], Lt=T-P               (text; decoder bos corner)
 *   i<P<=j   : relx[h][0]          i>=P>j : relx[h][1]    (decoder bos col / row)
 * causal=1 uses "tail-first" order: grid query i sees grid keys j<=i and all tail
 * keys; tail query i sees tail keys j<=i only.  dense_bias: optional fp32 [H,T,dense_ld]
 * (resized grids: ifseg_resized_rel_bias; dense_ld >= S is the row stride, 0 = S; a multiple of 4 with no rel_mode / causal
 * takes the seeded path: four 16-byte loads per 32-key block into the score accumulator, -inf entries = masked keys).
 * Requires P % 64 == 0 when rel_mode or causal. */
int ifseg_attn_fwd(const void* q, const void* k, const void* v, const void* pos_q, const void* pos_k,
                   void* out, float* lse, int B, int H, int T, int S, int ldq, int ldk, int ldv, int ldo,
                   int ldpq, int ldpk, long long q_bs, long long k_bs, long long v_bs, long long o_bs,
                   int rel_mode, int P, const int* gcode, int code_bias, int n2d, const float* rel2d,
                   const float* rel1d, const float* relx, int causal, const float* dense_bias,
                   const void* gain /* fp32 [H] or NULL */,
                   int grid_w /* token-grid width if gcode is the raster code y*(2w-1)+x, else 0; 32 enables
                                 the row-aligned bias lookup; any other multiple of 8 >= 32: seeds per group of 8 */,
                   int dense_ld /* row stride of dense_bias (0: S) */,
                   void* stream);

/* Backward of ifseg_attn_fwd (autograd of the same reference lines).  Launches
 *   delta[b,h,t] = sum_d dout*out;  a key-stationary dK/dV kernel that also
 *   histograms d(rel2d/rel1d/relx) in LDS;  a query-stationary dQ kernel.
 * `gain` [H] (fp32) is the per-head c_attn applied to `out` by the forward
 * (unify_multihead_attention.py:509-512); d(gain)[h] = sum_{b,t} delta / gain[h].
 * dq is scaled by dq_scale (the reference's q scaling), the abs-pos halves of
 * dQ_ext / dK_ext are written as bf16 per-batch partials dpos_q_part [B,T,H*64]
 * (scaled by dpq_scale) and dpos_k_part [B,S,H*64]; rel-table gradients as
 * per-workgroup partials [H][nparts][n] with nparts = B*ceil(S/128) (sum them with
 * ifseg_attn_bwd_reduce).  No atomics on global memory: results are deterministic. */
typedef struct ifseg_attn_bwd_args {
  const void *q, *k, *v, *pos_q, *pos_k, *out, *dout;
  const float* lse;
  float* delta;               /* workspace [B,H,T] */
  void *dq, *dk, *dv;
  void *dpos_q_part, *dpos_k_part; /* bf16 */
  int B, H, T, S;
  int ldq, ldk, ldv, ldpq, ldpk, ldout, lddo, lddq, lddk, lddv;
  long long q_bs, k_bs, v_bs, out_bs, do_bs, dq_bs, dk_bs, dv_bs;
  int rel_mode, P, code_bias, n2d, causal, nparts;
  const int* gcode;
  const float *rel2d, *rel1d, *relx;
  const void* gain; /* fp32 [H] or NULL */
  float *drel2d_part, *drel1d_part, *drelx_part;
  float dq_scale, dpq_scale;
  int grid_w; /* width of the token grid (0 if unknown); 32 enables the row-aligned bias-gradient reduction */
  int phases; /* 0 = everything; else a mask: 1 = delta, 2 = dK/dV kernel, 4 = dQ kernel.  The dQ kernel only needs
                 delta, so a caller may launch {1|2} and {4} on two streams (ordered by an event after delta): the two
                 kernels then fill each other's partially occupied last round of workgroups. */
  float* dgain_rows; /* optional fp32 [B,H,T], written by the dQ kernel: sum_j P_ij dP_ij = dout_i . (P v)_i -- the per-row terms of
                 d c_attn[h] without the division of delta by c_attn (0 / 0 at c_attn = 0): sum them instead of delta / gain */
} ifseg_attn_bwd_args;
#define IFSEG_ATTN_BWD_DELTA 1
#define IFSEG_ATTN_BWD_DKV 2
#define IFSEG_ATTN_BWD_DQ 4
int ifseg_attn_bwd(const ifseg_attn_bwd_args* args, void* stream);

/* Everything the partial outputs of ifseg_attn_bwd still need, in ONE launch (the autograd reductions behind
 * unify_multihead_attention.py:459-512 + the bias construction encoder_module.py:757-809 / decoder_module.py:553-631:
 * sum over the batch of the abs-pos operand gradients, sum over the workgroup partials of the rel-pos table gradients
 * followed by embedding_dense_backward into the bucket tables, and the gradient of c_attn):
 *   dpos_q_acc[T*C]  (=|+=) sum_b dpos_q_part[b]        dpos_k_acc[S*C] (=|+=) sum_b dpos_k_part[b]
 *   dgain[h] (bf16)  = sum_{b,t} delta[b,h,t] / gain[h]          (delta = rowsum(dO * O), O includes the gain)
 *   for each of up to 3 tables i:  acc_i[idx_i[j]][h] += sum_p part_i[h][p][j]   (idx < 0: no bucket)
 * Tables of up to 2048 entries (token offsets: several entries may share a bucket) are summed per bucket in entry order,
 * larger ones (the image / seg grids: entries and buckets one-to-one) by one atomic add per entry: bit-reproducible either way.
 * Replaces 2 + 4 + 2 per table launches of ifseg_reduce_parts / ifseg_rel_scatter_add / elementwise kernels. */
typedef struct ifseg_attn_reduce_args {
  int B, H, T, S, C, nparts, accumulate_pos;
  const void *dpos_q_part, *dpos_k_part;    /* bf16 [B,T,C], [B,S,C] (as ifseg_attn_bwd writes them) */
  float *dpos_q_acc, *dpos_k_acc;           /* [T,C], [S,C] */
  const float* delta;                       /* [B,H,T] */
  const float* gain;                        /* fp32 [H] */
  void* dgain;                              /* bf16 [H] or NULL */
  int ntab;
  const float* tab_part[3];                 /* [H,nparts,n_i] */
  const int* tab_idx[3];                    /* [n_i] bucket of entry j of the delta table */
  float* tab_acc[3];                        /* fp32 [n_bucket_i, H] accumulators */
  int tab_n[3];
  int tab_nbucket[3];                       /* rows of tab_acc_i */
} ifseg_attn_reduce_args;
int ifseg_attn_bwd_reduce(const ifseg_attn_reduce_args* args, void* stream);

/* ---- attention backward with the batch as a workgroup's inner dimension (csrc/attention_bi.hip) ----
 * The reference builds abs-pos + rel-pos bias ONCE per layer and broadcasts it over the batch
 * (encoder_module.py:757-771,790-809 with the expand at :317,791; decoder_module.py:553-558,603-627;
 * unify_multihead_attention.py:459-465).  These three entry points keep that structure:
 *
 * ifseg_attn_dense_bias: D[h][i][j] = pos_q[i].pos_k[j] + rel(i,j) as bf16 [H,Tp,Sp] (computed in fp32, rounded once; the
 *   reference holds this tensor in half precision under --fp16, unify_multihead_attention.py:464; Sp / Tp = S / T rounded up
 *   to 32; D 16-byte aligned); -inf where the causal mask ("tail-first" order of ifseg_attn_fwd) hides (i,j), for padded columns j >= S
 *   and padded rows i >= T.  Parameters only: built once per layer and step.  (Causal: 32 x 32 tiles no kernel's block
 *   schedule reaches are not written.)
 *   rel(i,j) as documented at ifseg_attn_fwd (rel_mode = 1), pos_q / pos_k may be NULL (no abs-pos term). */
int ifseg_attn_dense_bias(const void* pos_q, const void* pos_k, int ldpq, int ldpk, int H, int T, int S, int rel_mode,
                          int P, const int* gcode, int code_bias, int n2d, const float* rel2d, const float* rel1d,
                          const float* relx, int causal, void* D, int Sp, int Tp, void* stream);

/* ifseg_attn_bwd_bi: dq (x dq_scale), dk, dv of S_b = q_b k_b^T + D (autograd of unify_multihead_attention.py:459-512)
 *   and dbias[g][h][i][j] = sum over the batch elements 4g .. 4g+3 of dS_b[h][i][j] (bf16 [ceil(B/4), H, T, Sp]).
 *   One workgroup = (head, 64 rows, 4 batch elements): the bias tile is fetched once for the four, sum_b dS leaves the
 *   kernel once per tile.  delta [B,H,T] = rowsum(dout * out) must exist (ifseg_attn_bwd phase 1 or
 *   ifseg_gemm_nn_rowdot).  Causal launches skip 32-blocks above the diagonal: those entries of dbias are NOT written --
 *   the caller zero-fills the buffer once (the set of skipped blocks depends only on the shape).  No atomics. */
typedef struct ifseg_attn_bi_args {
  const void *q, *k, *v, *dout;
  const float *lse, *delta;
  const void* D;                /* bf16 [H, Tp, Sp] (ifseg_attn_dense_bias) */
  const void* gain;            /* fp32 [H] or NULL */
  void *dq, *dk, *dv, *dbias;
  int B, H, T, S, Sp, Tp;
  int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  long long q_bs, k_bs, v_bs, do_bs, dq_bs, dk_bs, dv_bs;
  int causal, P;               /* P: grid tokens (multiple of 64) when causal */
  float dq_scale;
  int phases;                  /* 0 = both; IFSEG_ATTN_BWD_DKV | IFSEG_ATTN_BWD_DQ */
  float* dgain_rows;           /* optional fp32 [B,H,T]: sum_j P_ij dP_ij = dout_i . (P v)_i, the per-row terms of d c_attn[h]
                                  (unify_multihead_attention.py:509-512) computed WITHOUT dividing delta by c_attn: exact at c_attn = 0 */
  void* out;                   /* ifseg_attn_fwd_bi: O [B,T,ldout] bf16 (x gain) */
  int ldout; long long out_bs;
  const int* kv_len;           /* optional device int32 [B]: key padding (unify_multihead_attention.py:477-489 with the suffix masks of
                                  encoder_module.py:730-752): keys j >= kv_len[b] are masked for batch element b in the forward and in
                                  all three gradients; NULL = no padding.  kv_len[b] >= 1 */
  float drop_p;                /* attention dropout (unify_multihead_attention.py:498: dropout_module(attn_weights)): P o keep / (1 - p)
                                  between the softmax and P V, keep ~ Bernoulli(1 - p) from a counter-based hash of
                                  (drop_seed + *drop_seed_add, b, h, query, key) that forward and the three gradients regenerate
                                  (ifseg_attn_dropout_mask writes it out); 0 = off.  `delta` must come from the dropped output. */
  unsigned long long drop_seed;
  const unsigned long long* drop_seed_add;   /* device word or NULL (as ifseg_drop_args.seed_add) */
} ifseg_attn_bi_args;
int ifseg_attn_bwd_bi(const ifseg_attn_bi_args* args, void* stream);
/* ifseg_attn_fwd_bi: the forward of the same formulation -- out = gain softmax_fp32(q k^T + D) v, `lse` (written) as
 * ifseg_attn_fwd's (log2 units); reads q, k, v, D, gain, causal / P (tile schedule only: the mask is inside D).  A workgroup
 * holds four batch elements and fetches each 32 x 32 bias tile once for them (unify_multihead_attention.py:459-512). */
int ifseg_attn_fwd_bi(const ifseg_attn_bi_args* args, void* stream);
/* keep[b, h, i, j] in {0, 1} (uint8 [B, H, T, S]): the attention-dropout mask of the kernels above for these seeds */
int ifseg_attn_dropout_mask(unsigned char* keep, int B, int H, int T, int S, float p, unsigned long long seed,
                            const unsigned long long* seed_add, void* stream);

/* ifseg_attn_dbias_grads: everything downstream of dbias (two launches: operand gradients, tables) (the autograd of the bias construction,
 *   encoder_module.py:757-771,790-809 / decoder_module.py:553-558,603-627), with dB = sum_g dbias[g]:
 *     dpos_q_acc[i][h*64+c] (=|+=) dpq_scale * sum_j dB[h][i][j] pos_k[j][h*64+c]       (fp32 [T,C])
 *     dpos_k_acc[j][h*64+c] (=|+=)             sum_i dB[h][i][j] pos_q[i][h*64+c]       (fp32 [S,C])
 *     drel2d[h][p][(dy+gh-1)(2gw-1) + dx+gw-1] = sum of dB over grid pairs with (y_i-y_j, x_i-x_j) = (dy, dx)  (raster grid gh x gw = P, gw % 8 == 0)
 *     drel1d[h][p][(i-j)+Lt-1] = sum over tail pairs;  drelx[h][p][0] = sum_{i<P<=j<S} dB,  drelx[h][p][1] = sum_{j<P<=i<T} dB
 *   as NP = ifseg_attn_dbias_nparts() partial tables per head (part p = the rows i = p mod NP): they feed
 *   ifseg_attn_bwd_reduce with nparts = NP.  pos_q == NULL skips the operand gradients, drel2d == NULL the tables.
 *   Fixed summation order.  Any ng: the slabs are taken two per launch pair, later pairs add to the first's results. */
typedef struct ifseg_attn_dbias_args {
  const void* dbias;           /* bf16 [ng][H][T][Sp] */
  int ng, H, T, S, Sp, C;
  const void *pos_q, *pos_k;   /* bf16 [T,ldpq], [S,ldpk] */
  int ldpq, ldpk;
  float *dpos_q_acc, *dpos_k_acc;
  int accumulate_pos;
  float dpq_scale;
  int P, grid_h, grid_w;
  float *drel2d, *drel1d, *drelx;
  int causal;                  /* dbias comes from a causal launch of ifseg_attn_bwd_bi: the blocks it skipped are not read */
} ifseg_attn_dbias_args;
int ifseg_attn_dbias_grads(const ifseg_attn_dbias_args* args, void* stream);
int ifseg_attn_dbias_nparts(void);

/* -------------------------------------------------------------- row ops */
/* Row addressing used below: logical row r lives at element offset
 *   (r / rpb) * bs + (r % rpb) * ld      (rpb = rows per batch segment; rpb<=0: r*ld)
 * which lets a kernel read / write a token range of a [B, T, C] buffer in place
 * (the reference's torch.cat of image|text tokens, encoder_module.py:427, and of
 * bos|patch tokens, decoder_module.py:537). */

/* y = [resid +] LayerNorm(act(x)) * gamma + beta, act = identity | GELU(fp32).
 * mean/rstd (fp32 [rows]) are saved for the backward when non-NULL.
 * Replaces fairseq LayerNorm (modules/layer_norm.py:30-35) at the sites
 * unify_transformer_layer.py:258,270,278,282,465,513,522,546,554,559 and
 * encoder_module.py:408,423,757-759,829 / decoder_module.py:344,576,668, fused with
 * activation_fn GELU (modules/gelu.py:24-25) and residual_connection (:196). */
/* Optional fused dropout + DropPath of the LayerNorm output (forward) / of dy (backward): the mask is the one
 * ifseg_dropout generates for the same (seed, logical row, column), so stand-alone and fused calls can be mixed
 * (FairseqDropout after attn_ln / cross_attn_ln / the embedding LayerNorms, drop_path in residual_connection:
 * unify_transformer_layer.py:19-35,196).  NULL = no dropout. */
typedef struct ifseg_drop_args {
  float p;                       /* drop probability in [0, 1) */
  unsigned long long seed;
  const float* drop_path_scale;  /* [batch] fp32 keep/(1-rate) per sample, or NULL */
  int rows_per_batch;            /* logical rows per sample (indexes drop_path_scale) */
  const unsigned long long* seed_add; /* device word added to `seed`, or NULL: the per-update part of the seed lives in
                                       * device memory so that a HIP-graph-captured step draws new masks on every replay */
} ifseg_drop_args;
/* `act_gelu` / `flags` of the LayerNorm entry points: IFSEG_LN_GELU = the input goes through GELU first (ffn_layernorm(gelu(fc1)));
 * IFSEG_LN_PARAMS_F32 = gamma / beta (and gamma2 / beta2) point at fp32 values (the optimizer's master copy) instead of bf16. */
#define IFSEG_LN_GELU 1
#define IFSEG_LN_PARAMS_F32 2
int ifseg_ln_fwd(const void* x, const void* gamma, const void* beta, const void* resid, void* y, float* mean,
                 float* rstd, int rows, int C, float eps, int act_gelu, int rpb, long long x_bs, int ldx,
                 long long y_bs, int ldy, long long r_bs, int ldr, const ifseg_drop_args* drop, void* stream);
/* Two LayerNorms of one residual-stream row in one pass (post-LN of a block + pre-LN of the next,
 * unify_transformer_layer.py:256-292,463-568): y = [resid +] drop(LN(x; gamma, beta)), y2 = LN(y as stored in bf16;
 * gamma2, beta2).  Bit-identical to ifseg_ln_fwd followed by ifseg_ln_fwd on y.  gamma == beta == NULL makes the
 * first stage the identity: y = [resid +] drop(x) (the dropout + residual after fc2, identical to ifseg_dropout),
 * y2 = LN(y) -- the pre-LN of the next layer. */
int ifseg_ln_fwd_pair(const void* x, const void* gamma, const void* beta, const void* resid, void* y, float* mean,
                      float* rstd, const void* gamma2, const void* beta2, void* y2, float* mean2, float* rstd2, int rows,
                      int C, float eps, int flags, int rpb, long long x_bs, int ldx, long long y_bs, int ldy, long long r_bs, int ldr,
                      long long y2_bs, int ldy2, const ifseg_drop_args* drop, void* stream);
/* dx = [dx_add +] d/dx of the above (dy is first masked like the forward output when `drop` is given);
 * per-block partials of dgamma / dbeta are written to dgamma_part / dbeta_part [nblocks][C]
 * (sum with ifseg_reduce_parts). */
int ifseg_ln_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                 const void* dx_add, void* dx, float* dgamma_part, float* dbeta_part, int nblocks, int rows,
                 int C, int act_gelu, int rpb, long long dy_bs, int lddy, long long x_bs, int ldx,
                 long long dx_bs, int lddx, long long add_bs, int ldadd, const ifseg_drop_args* drop, void* stream);
/* ifseg_ln_bwd (C <= 1024, no GELU) with a second output dx2 = drop2(dx as stored in bf16): the pre-LN backward that
 * closes a block of the backward and the adjoint of the dropout + DropPath after fc2 (unify_transformer_layer.py:288-292,
 * 564-568) that opens the next one, in one launch.  Bit-identical to ifseg_ln_bwd followed by ifseg_dropout. */
int ifseg_ln_bwd_drop(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                      const void* dx_add, void* dx, float* dgamma_part, float* dbeta_part, void* dx2, int nblocks, int rows,
                      int C, int flags, int rpb, long long dy_bs, int lddy, long long x_bs, int ldx, long long dx_bs, int lddx,
                      long long add_bs, int ldadd, long long dx2_bs, int lddx2, const ifseg_drop_args* drop2, void* stream);
/* Input validation in one launch, the verdict (0 / 1) written to PINNED host memory in stream order (read it after an event):
 * mode 0: any int64 x[i] == value; mode 1: rows of row_len int64: `value` entries that are not a suffix of their row (fairseq pads on
 * the right: encoder_module.py:730-752); mode 2: any byte x[i] == 0 (the bool `patch_masks`); mode 3: *x (int32 device flag) != 0,
 * and the flag is cleared.  Replaces compare + reduce + cast + copy torch kernels on the main queue between two steps. */
int ifseg_check_inputs(const void* x, long long n, int mode, long long value, long long row_len,
                       unsigned char* verdict_host_pinned, void* stream);
/* fp32 master copy <- bf16 arena wherever bf16(master[i]) != p16[i] (an optimizer outside this library stepped the bf16
 * parameters: fp16_optimizer.py:198-222 writes the model copy); agreeing entries keep their fp32 value. */
int ifseg_sync_master(float* master, const void* p16, long long n, void* stream);
/* out[o][i] (+)= scale * sum_p in[o][p][i]   (fp32 in; fp32 or bf16 out) */
int ifseg_reduce_parts(const float* in, void* out, int outer, int parts, long long n, int accumulate,
                       int out_bf16, float scale, void* stream);
/* Up to 16 such reductions in one launch (the LayerNorm dgamma / dbeta partials of one layer block). */
typedef struct ifseg_reduce_task {
  const float* in; void* out;
  int outer, parts; long long n;
  int accumulate, out_bf16; float scale;
} ifseg_reduce_task;
int ifseg_reduce_parts_multi(int ntask, const ifseg_reduce_task* tasks, void* stream);
/* part[k][n] = column sums of the k-th row slab of x[M,N] (bias gradients; autograd of
 * the bias add in F.linear). */
int ifseg_colsum_bf16(const void* x, float* part, int nblk_rows, int M, int N, int rpb, long long x_bs, int ldx,
                      void* stream);
/* out[r] = table[ids[r]] + add  (embed_tokens + type_embedding, encoder_module.py:400-406) */
/* gw [N, C] (bf16, a key projection's weight gradient dK^T x) -= db [N] (x) mean_rows(x) [C]: sum_j dK_j = 0 in exact arithmetic
 * (softmax shift invariance; the reference's k_proj.bias gradient is float noise), so this removes only the product of the
 * spurious bf16 column sum of dK with the token-common component of x.  xmean [C] fp32: the column means of x
 * (ifseg_colsum_bf16 + ifseg_reduce_parts).  Autograd of unify_multihead_attention.py:327-346. */
int ifseg_kproj_common_mode(void* gw, const void* db, const float* xmean, int N, int C, void* stream);
int ifseg_embed_rows(const void* table, const long long* ids, const void* add, void* out, int n, int C, int rpb,
                     long long o_bs, int ldo, void* stream);
/* Backward of the embedding look-ups above into the token table's gradient (trainable token embeddings:
 * --freeze-encoder-embedding / --freeze-decoder-embedding false, unify_transformer.py:362-371; nn.Embedding /
 * nn.EmbeddingBag(mean) backward).  m entries sorted by token id (stable); entry j adds weight[j] (NULL: 1) x src[src_row[j], :]
 * (bf16 [*, C] contiguous) to row sorted_ids[j] of table_grad (bf16 [V, C], accumulated in place); ids outside [0, V) and
 * skip_id (the padding index, -1: none) are skipped.  Deterministic: the first entry of a run sums the run in order. */
int ifseg_rows_segment_sum(const void* src, const long long* src_row, const float* weight, const long long* sorted_ids,
                           void* table_grad, long long m, int C, long long V, long long skip_id, void* stream);
/* Image-free patch embeddings: out[b,p,:] = mean of table rows ids[b, ends[b,p-1]:ends[b,p]] + add
 * (nn.EmbeddingBag(mode='mean') sharing embed_tokens.weight, encoder_module.py:147-148,529-538, + the image
 * type embedding :589-591).  ids int64 [B,maxlen] (row-padded), ends int64 [B,P] per-sample cumulative bag ends
 * exactly as the collater produces them (segmentation_dataset.py:327-328,99-100); out bf16 rows addressed like
 * ifseg_embed_rows. */
int ifseg_embed_bag_mean(const void* table, const long long* ids, const long long* ends, const void* add, void* out,
                         int B, int P, int C, int maxlen, int rpb, long long o_bs, int ldo, void* stream);
int ifseg_cast_f32_bf16(const float* in, void* out, long long n, float scale, void* stream);
int ifseg_add_bf16(const void* a, const void* b, void* out, long long n, void* stream);
/* NCHW (fp32 / bf16) image -> NHWC bf16 with channels zero-padded to Cpad */
int ifseg_nchw_to_nhwc_bf16(const void* in, int in_is_f32, void* out, int B, int C, int H, int W, int Cpad,
                            void* stream);

/* out[h][i] = table[idx[i]][h] (fp32; idx<0 -> 0): turns a rel-pos bucket table
 * [num_buckets, H] into the per-head delta table the attention kernels index
 * (F.embedding(rp_bucket, table) of encoder_module.py:313-331, decoder_module.py:327-333).
 * ifseg_rel_scatter_add is its adjoint (embedding_dense_backward) into an fp32 buffer. */
int ifseg_rel_gather(const void* table, const int* idx, float* out, int n, int H, void* stream);
/* the same gather for L <= 16 tables of one shape in one launch (all layers' tables at the start of a forward):
 * tables = host array of L device pointers, out fp32 [L][H][n] */
int ifseg_rel_gather_multi(const void* const* tables, int L, const int* idx, float* out, int n, int H, void* stream);
int ifseg_rel_scatter_add(const float* d, const int* idx, float* acc, int n, int H, void* stream);

/* Variable-aspect evaluation (criterions/seg_criterion.py:194-217 feeds images at their native aspect ratio): the reference
 * resizes its [H, P0, P0] relative-position bias with two bilinear interpolations per layer when the feature grid (h, w)
 * differs from the trained (oh, ow) one (encoder_module.py:802-808, decoder_module.py:603-627).  out fp32 [H, T, T],
 * T = h*w + Lt, internal order [grid | tail]: grid x grid = the doubly resized bias, evaluated as 4 x 4 taps of the ORIGINAL
 * delta table table2d [H, (2oh-1)(2ow-1)] (ifseg_rel_gather on the original geometry); tail x tail = rel1d [H, 2Lt-1]
 * (NULL: 0); grid x tail / tail x grid = relx [H, 2] (NULL: 0) -- the decoder's bos column / row, which the reference's
 * resize passes through.  causal != 0 (decoder_module.py:592-600, buffered_future_mask in the internal order with the tail =
 * bos first): -inf where the key lies after the query. */
int ifseg_resized_rel_bias(float* out, const float* table2d, const float* rel1d, const float* relx, int H, int h, int w,
                           int oh, int ow, int Lt, int causal, int ld /* row stride of out, >= T (0: T) */, void* stream);
/* dst bf16 [h*w, C] = bilinear resize (align_corners = False, fp32 arithmetic, one rounding) of the oh x ow grid of rows
 * src[(y * src_stride + x + src_off) * ld_src + c]: the position-embedding tables of encoder_module.py:360-368 and
 * decoder_module.py:541-548 on a resized grid */
int ifseg_resize_rows_bilinear(const void* src, void* dst, int C, int h, int w, int oh, int ow, int src_stride, int src_off,
                               int ld_src, void* stream);

/* out = [resid +] drop_path_scale[row / rows_per_batch] * keep * x / (1 - p) on [rows, C] bf16 (row addressing as above);
 * keep ~ Bernoulli(1-p) from a counter-based hash of (seed, element): calling it again with x = dy,
 * resid = NULL is the backward.  FairseqDropout (fairseq_dropout.py:23-27) + drop_path
 * (unify_transformer_layer.py:19-35, residual_connection :196). */
int ifseg_dropout(const void* x, const void* resid, void* out, long long rows, int C, float p,
                  unsigned long long seed, const float* drop_path_scale, int rows_per_batch, int rpb, long long x_bs,
                  int ldx, long long r_bs, int ldr, long long o_bs, int ldo, const unsigned long long* seed_add, void* stream);
/* out[i] = keep(i) ? x[i] : fill on n contiguous bf16 elements (n % 8 == 0), ifseg_dropout's mask, NO rescaling: the
 * activation dropout between GELU and the FFN LayerNorm (unify_transformer_layer.py:280,556) applied to the pre-activation
 * with fill = -30 (gelu and gelu' are exactly 0 there); the 1 / (1 - p) cancels in the LayerNorm: pass eps (1 - p)^2 to
 * ifseg_ln_fwd (csrc/rowops.hip). */
int ifseg_dropout_fill(const void* x, void* out, long long n, float p, unsigned long long seed,
                       const unsigned long long* seed_add, float fill, void* stream);
/* DropPath keep masks (drop_path, unify_transformer_layer.py:19-35): out[i][b] = Bernoulli(keep[i]) / keep[i] for residual
 * branch i < n and sample b < B, from the counter-based generator of ifseg_dropout (seed + *seed_add). */
int ifseg_droppath_scale(float* out, const float* keep, int n, int B, unsigned long long seed,
                         const unsigned long long* seed_add, void* stream);

/* ------------------------------------------------------------ ResNet stem */
/* conv1 7x7/2 (3->64) + folded FrozenBN + ReLU on an NHWC(4) bf16 image; w fp32
 * [7][7][3][64] with the BN scale folded, shift fp32 [64] (resnet.py:215-218). */
int ifseg_stem_conv7x7(const void* in_nhwc4, const float* w, const float* shift, void* out, int B, int H, int W,
                       void* stream);
/* The same on the matrix cores: wt bf16 [3][64][224], wt[0][c][ky*32 + kx*4 + ci] = bf16(w[ky][kx][ci][c]) (BN scale folded; the
 * entries with kx = 7 or ci = 3 are zero), wt[1] = bf16(w - wt[0]), wt[2] = bf16(w - wt[0] - wt[1]): implicit GEMM,
 * K = 7 x 8 x 4, the fp32 weights as three bf16 terms (resnet.py:215-218). */
int ifseg_stem_conv7x7_mfma(const void* in_nhwc4, const void* wt, const float* shift, void* out, int B, int H, int W,
                            void* stream);
/* MaxPool2d(3, 2, 1) on NHWC bf16 (resnet.py:219). */
int ifseg_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------- criterion */
/* Fused segmentation loss: bilinear x16 upsample of the per-patch logits
 * (logits[:, :P] viewed as [hp, wp]; align_corners=False), masked mean cross entropy, its
 * gradient w.r.t. the logits, argmax and the per-class area histograms, replacing
 * upsample_logits + F.cross_entropy + compute_metric (criterions/seg_criterion.py:237-244,
 * :269-347, :349-362) and their autograd.  Step 1 (one workgroup per 16x16 pixel tile) writes
 *   tile_partial [B*hp*wp][3][3][nseg]  gradient of the tile w.r.t. its 3x3 stencil cells
 *   stats_part   [B*hp*wp][2 + 3*nseg]  loss sum, valid-pixel count, intersect|pred|label hist
 * the caller sums stats_part over tiles (ifseg_reduce_parts) into stats[2+3*nseg]; step 2
 * gathers dlogits (bf16 [B, P+1, ldl], eos row and padding columns zero) scaled by 1/valid and
 * writes the mean loss.  target: int64 [B, H*W+1] dictionary ids; a pixel is ignored when its
 * target is pad, eos or <seg_nseg>.  label_smoothing in [0, 1]: F.cross_entropy's epsilon (seg_criterion.py:265; 0 = plain CE,
 * bit-identical to ABI <= 14).  Requires H == 16*hp, W == 16*wp, 1 <= nseg <= 512. */
int ifseg_seg_loss_tiles(const void* logits, int ldl, long long logits_bs, const long long* target,
                         long long target_bs, int B, int hp, int wp, int H, int W, int nseg,
                         long long seg_id_offset, long long pad_id, long long eos_id, float* tile_partial,
                         float* stats_part, int* bad_label /* device flag, set to 1 when a target is neither a class nor
                         pad / eos / ignore (F.cross_entropy would raise); may be NULL */, float label_smoothing, void* stream);
int ifseg_seg_loss_gather(const float* tile_partial, const float* stats, void* dlogits, int ldl,
                          long long dlogits_bs, int B, int hp, int wp, int nseg, float* loss_out, void* stream);

/* -------------------------------------------------------------- optimizer */
/* ---- eval-time post-processing of the criterion (BASELINE config 5) ----------------------
 * top-k neighbour smoothing (criterions/seg_criterion.py:197-213):
 *   f = ifseg_l2norm_rows_bf16(trunk features [B*P, D]);  sim = f f^T (ifseg_gemm_bf16 NT, fp32 out, per batch);
 *   idx = ifseg_topk_rows_f32(sim, k);  prob = ifseg_softmax_rows(logits / temperature);
 *   iters x: prob = ifseg_gather_mean(prob, idx)
 * metrics at the original image resolution (:289-347): ifseg_seg_eval resizes the [hp, wp, n] fp32 score grid
 * bilinearly (align_corners=False, any ratio) to [h, w], takes the argmax and accumulates hist[3][n] =
 * {intersect, predicted, label} pixel counts (uint64, caller zeroes them) and per-block {CE sum, pixel count}
 * partials loss_part[nblocks][2], nblocks = ceil(h*w / 256).  target: int64 [h*w] dictionary ids
 * (seg_id_offset + class; anything outside [0, n) after the offset is ignored). */
int ifseg_l2norm_rows_bf16(const void* x, void* out, int rows, int D, void* stream);
int ifseg_topk_rows_f32(const float* sim, int* idx, int rows, int N, int k /* <= 8 */, void* stream);
int ifseg_softmax_rows(const void* logits /* bf16 */, long long batch_stride, int rows_per_batch, int ld, float* prob,
                       int rows, int n, float inv_temperature, int do_softmax /* 0: fp32 copy only */, void* stream);
int ifseg_gather_mean(const float* in, const int* idx, float* out, int B, int P, int n, int k, void* stream);
int ifseg_seg_eval(const float* scores, int hp, int wp, int n, const long long* target, int h, int w,
                   long long seg_id_offset, unsigned long long* hist, float* loss_part, int nblocks, void* stream);

/* sum of squares of a bf16 gradient arena -> out_sumsq[0] (device). */
int ifseg_grad_sumsq_bf16(const void* g, long long n, float* workspace, float* out_sumsq, void* stream);
/* Fused grad scaling + clip-by-global-norm + Adam with decoupled weight decay over a
 * flat arena; writes the bf16 model copy.  Mirrors trainer.py:874-907 +
 * fairseq/optim/adam.py:158-240 + fp16_optimizer.py (fp32 masters).  A non-finite *sumsq skips the whole update and
 * sets overflow[0] = 1 (device int, may be NULL): trainer.py:895-904 raises FloatingPointError there. */
int ifseg_adam_step(float* p32, const void* g, float* m, float* v, void* p16, long long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, float max_norm,
                    const float* sumsq, int* overflow, const float* hyper, void* stream);
/* `hyper` (device, may be NULL): {lr, 1 - beta1^step, 1 - beta2^step, grad_scale} override the by-value arguments --
 * a captured training step (HIP graph) is replayed with the schedule's current values. */

/* ------------------------------------------------- dense-CRF post-processing (crf.py:19-37) */
/* Mean-field inference of the reference's DenseCRF2D (pydensecrf, third party, absent: parity unpinned) with the EXACT
 * dense kernels instead of the permutohedral-lattice approximation.  All tensors class-major ([C][N], N = H*W pixels).
 * ifseg_crf_bilateral: out[c][i] = sum_j gx[|xi-xj|] gy[|yi-yj|] exp2(-|f_i - f_j|^2) qn[c][j]; feat4 [N][4] fp32 =
 *   (pre-scaled r, g, b, bit pattern x | y << 16), qn bf16 [Cp][ldq] (Cp % 32 == 0, ldq % 64 == 0, zero padded), out fp32
 *   [Cp][N].  ifseg_crf_spatial: the sxy = 1 Gaussian as a (2R+1)^2 stencil on fp32 [C][N].
 * ifseg_crf_update: Q = softmax_c(log clip(prob, 1e-5, 1) + wpos npos mpos + wbi nbi mbi) -> Q, qpos = npos Q (fp32),
 *   qbi = nbi Q (bf16 [Cp][ldq]); mpos / mbi / npos / nbi may be NULL.  ifseg_crf_norm: n = rsqrt(k1 + 1e-20). */
int ifseg_crf_bilateral(const float* feat4, const float* gx, const float* gy, const void* qn, int ldq, float* out, int Cp,
                        int H, int W, void* stream);
int ifseg_crf_spatial(const float* qn, const float* g, int R, float* out, int C, int H, int W, void* stream);
int ifseg_crf_update(const float* prob, const float* mpos, const float* mbi, const float* npos, const float* nbi, float wpos,
                     float wbi, float* Q, float* qpos, void* qbi, int ldq, int C, int N, void* stream);
int ifseg_crf_norm(const float* k1, float* n, int N, void* stream);

/* -------------------------------------------------------------- profiling */
/* Per-kernel-family timing with HIP events recorded on the launch stream (bench.py's
 * roofline object).  kinds: 0 GEMM_NT, 1 GEMM_NN, 2 GEMM_TN, 3 CONV, 4 ATTN_FWD,
 * 5 ATTN_BWD_DKV, 6 ATTN_BWD_DQ, 7 LN_FWD, 8 LN_BWD.  ifseg_prof_read returns the summed
 * kernel time and the summed ALGORITHMIC flops / bytes of the recorded launches. */
int ifseg_prof_enable(unsigned mask);
int ifseg_prof_stride(int stride); /* time only every stride-th launch of an enabled family (default 1) */
int ifseg_prof_reset(void);
int ifseg_prof_read(int kind, double* ms, double* flops, double* bytes, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* IFSEG_HIP_H */
