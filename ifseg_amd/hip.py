"""ctypes binding of libifseg_hip.so (C ABI: include/ifseg_hip.h).

PyTorch tensors are used only as owners of device memory and for the current
HIP stream; every call passes raw device pointers + sizes.  There is NO
fallback: a missing library or a non-zero return code raises RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libifseg_hip.so")
_lib = None

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
GEMM_RELU, GEMM_OUT_F32, GEMM_ACCUMULATE = 1, 2, 4

c_void_p, c_int, c_float, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ifseg_amd: %s not found -- build it with `python -m ifseg_amd.build` "
                "(there is no CPU/PyTorch fallback for the HIP path)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ifseg_abi_version.restype = c_int
        if _lib.ifseg_abi_version() != 1:
            raise RuntimeError("ifseg_amd: ABI version mismatch")
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("ifseg_amd HIP call %s failed with code %d" % (what, rc))


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor required"
    return c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf(t):
    assert t.dtype == torch.bfloat16, t.dtype
    return t


# --------------------------------------------------------------------------- GEMM
def gemm(layout, A, B, C, M, N, K, lda, ldb, ldc, bias=None, alpha=1.0, alpha_ncols=-1, resid=None, ldr=0,
         flags=0, batch=1, sA=0, sB=0, sC=0, sR=0):
    rc = lib().ifseg_gemm_bf16(c_int(layout), _ptr(A), _ptr(B), _ptr(C), c_int(M), c_int(N), c_int(K),
                               c_int(lda), c_int(ldb), c_int(ldc), _ptr(bias), c_float(alpha), c_int(alpha_ncols),
                               _ptr(resid), c_int(ldr), c_int(flags), c_int(batch), c_ll(sA), c_ll(sB), c_ll(sC),
                               c_ll(sR), _stream())
    _check(rc, "gemm")


def linear_fwd(x, w, bias=None, out=None, alpha=1.0, alpha_ncols=-1, resid=None):
    """out[M,N] = ((x[M,K] @ w[N,K]^T + bias) * alpha) + resid   (bf16)"""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    gemm(GEMM_NT, _bf(x), _bf(w), out, M, N, K, x.stride(0), w.stride(0), out.stride(0), bias, alpha, alpha_ncols,
         resid, resid.stride(0) if resid is not None else 0)
    return out


def linear_dx(dy, w, out=None, resid=None, accumulate=False):
    """dx[M,K] = dy[M,N] @ w[N,K]  (+ resid)"""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=torch.bfloat16, device=dy.device)
    gemm(GEMM_NN, _bf(dy), _bf(w), out, M, K, N, dy.stride(0), w.stride(0), out.stride(0), None, 1.0, -1, resid,
         resid.stride(0) if resid is not None else 0, GEMM_ACCUMULATE if accumulate else 0)
    return out


def linear_dw(dy, x, out, accumulate=False):
    """dw[N,K] = dy[M,N]^T @ x[M,K]   (out bf16 or fp32 decided by out.dtype)"""
    M, N = dy.shape
    K = x.shape[1]
    flags = (GEMM_OUT_F32 if out.dtype == torch.float32 else 0) | (GEMM_ACCUMULATE if accumulate else 0)
    gemm(GEMM_TN, _bf(dy), _bf(x), out, N, K, M, dy.stride(0), x.stride(0), out.stride(0), flags=flags)
    return out


def conv2d_nhwc(x, w, shift, resid, out, B, H, W, Cin, Cout, KH, KW, stride, pad, relu):
    rc = lib().ifseg_conv2d_nhwc_bf16(_ptr(x), _ptr(w), _ptr(shift), _ptr(resid), _ptr(out), c_int(B), c_int(H),
                                      c_int(W), c_int(Cin), c_int(Cout), c_int(KH), c_int(KW), c_int(stride),
                                      c_int(pad), c_int(1 if relu else 0), _stream())
    _check(rc, "conv2d_nhwc")
    return out


# ---------------------------------------------------------------------- attention
class RelBias:
    """Per-(layer) rel-pos bias operands of the attention kernels (all device tensors)."""

    def __init__(self, P, gcode, code_bias, rel2d, rel1d, relx):
        self.P, self.gcode, self.code_bias = P, gcode, code_bias
        self.rel2d, self.rel1d, self.relx = rel2d, rel1d, relx   # fp32 [H,n2d], [H,2Lt-1], [H,2]


def attn_fwd(q, k, v, pos_q, pos_k, out, lse, B, H, T, S, rel=None, causal=False, P=None, dense_bias=None):
    """q/k/v/out: [B, T|S, *] bf16 row-strided views (head h at cols h*64..); pos_q/pos_k: [T|S, H*64]."""
    L = lib()
    if rel is not None:
        P = rel.P
    if P is None:
        P = S
    rc = L.ifseg_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(pos_q), _ptr(pos_k), _ptr(out), _ptr(lse), c_int(B),
                          c_int(H), c_int(T), c_int(S), c_int(q.stride(1)), c_int(k.stride(1)), c_int(v.stride(1)),
                          c_int(out.stride(1)), c_int(pos_q.stride(0) if pos_q is not None else 0),
                          c_int(pos_k.stride(0) if pos_k is not None else 0), c_ll(q.stride(0)), c_ll(k.stride(0)),
                          c_ll(v.stride(0)), c_ll(out.stride(0)), c_int(1 if rel is not None else 0), c_int(P),
                          _ptr(rel.gcode) if rel is not None else None, c_int(rel.code_bias if rel is not None else 0),
                          c_int(rel.rel2d.shape[1] if rel is not None else 0),
                          _ptr(rel.rel2d) if rel is not None else None, _ptr(rel.rel1d) if rel is not None else None,
                          _ptr(rel.relx) if rel is not None else None, c_int(1 if causal else 0), _ptr(dense_bias),
                          _stream())
    _check(rc, "attn_fwd")
    return out
