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

#define IFSEG_ABI_VERSION 1
#define IFSEG_ERR_BAD_SHAPE (-2)
#define IFSEG_ERR_BAD_ARG (-3)

int ifseg_abi_version(void);

/* ------------------------------------------------------------------ GEMM */
#define IFSEG_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T   (F.linear forward)        */
#define IFSEG_GEMM_NN 1 /* C[M,N] = A[M,K] . B[K,N]     (dX = dY . W)             */
#define IFSEG_GEMM_TN 2 /* C[M,N] = A[K,M]^T . B[K,N]   (dW = dY^T . X)           */
#define IFSEG_GEMM_RELU 1
#define IFSEG_GEMM_OUT_F32 2
#define IFSEG_GEMM_ACCUMULATE 4 /* C += result */

/* bf16 MFMA GEMM, fp32 accumulate, epilogue
 *   C = ((A.B + bias[n]) * alpha[for n < alpha_ncols]) + resid[m,n]  (+relu)
 * Replaces nn.Linear / F.linear and their autograd (addmm / mm) as used by
 * q/k/v/out_proj (unify_multihead_attention.py:327-346,513), fc1/fc2
 * (unify_transformer_layer.py:279-283,556-560), image_proj
 * (encoder_module.py:416), pos_{q,k}_linear (encoder_module.py:765-770,
 * decoder_module.py:350-363); `alpha` carries the reference's `q *= scaling`
 * (unify_multihead_attention.py:346) and `* pos_scaling`; `resid` the
 * residual_connection (unify_transformer_layer.py:196,289).
 * N, lda, ldb must be multiples of 8 (16-byte rows); bias/resid bf16. */
int ifseg_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                    int ldb, int ldc, const void* bias, float alpha, int alpha_ncols,
                    const void* resid, int ldr, int flags, int batch, long long strideA,
                    long long strideB, long long strideC, long long strideR, void* stream);

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
 * lse: fp32 [B,H,T].  Token order: grid tokens [0,P) then tail tokens [P,T)
 * (decoder: bos is moved to the end by the caller).  rel_mode=1:
 *   i,j <  P : rel2d[h][gcode[i]-gcode[j]+code_bias]     (image / seg grid)
 *   i,j >= P : rel1d[h][(i-j)+Lt-1], Lt=T-P               (text; decoder bos corner)
 *   i<P<=j   : relx[h][0]          i>=P>j : relx[h][1]    (decoder bos col / row)
 * causal=1 uses "tail-first" order: grid query i sees grid keys j<=i and all tail
 * keys; tail query i sees tail keys j<=i only.  dense_bias: optional fp32 [H,T,S]
 * (slow path for resized grids).  Requires P % 64 == 0 when rel_mode or causal. */
int ifseg_attn_fwd(const void* q, const void* k, const void* v, const void* pos_q, const void* pos_k,
                   void* out, float* lse, int B, int H, int T, int S, int ldq, int ldk, int ldv, int ldo,
                   int ldpq, int ldpk, long long q_bs, long long k_bs, long long v_bs, long long o_bs,
                   int rel_mode, int P, const int* gcode, int code_bias, int n2d, const float* rel2d,
                   const float* rel1d, const float* relx, int causal, const float* dense_bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IFSEG_HIP_H */
