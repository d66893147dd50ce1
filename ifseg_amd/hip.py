"""ctypes binding of libifseg_hip.so (C ABI: include/ifseg_hip.h).

PyTorch tensors are used only as owners of device memory and for the current
HIP stream; every call passes raw device pointers + sizes.  There is NO
fallback: a missing library or a non-zero return code raises RuntimeError.
"""
import ctypes
import os

import torch

from . import lab

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IFSEG_LIB", os.path.join(_HERE, "lib", "libifseg_hip.so"))
_lib = None

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
GEMM_RELU, GEMM_OUT_F32, GEMM_ACCUMULATE = 1, 2, 4

c_void_p, c_int, c_float, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong


ABI_VERSION = 18          # == IFSEG_ABI_VERSION of include/ifseg_hip.h (checked at load time and by __graft_entry__.build)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ifseg_amd: %s not found -- build it with `python -m ifseg_amd.build` "
                "(there is no CPU/PyTorch fallback for the HIP path)" % LIB_PATH)
        # bind to the HIP runtime PyTorch uses (one runtime per process; see ifseg_amd/build.py)
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if not os.path.exists(rt):
            rt = "libamdhip64.so"
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ifseg_abi_version.restype = c_int
        if _lib.ifseg_abi_version() != ABI_VERSION:
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


_stream_handle = None     # set by the engine for the duration of a forward / backward (saves a torch lookup per launch)


def set_stream(handle):
    """Pin the HIP stream every following launch goes to (raw hipStream_t as int), or None to follow
    torch.cuda.current_stream() again.  Returns the previous setting."""
    global _stream_handle
    prev = _stream_handle
    _stream_handle = handle
    return prev


def _stream():
    if _stream_handle is not None:
        return c_void_p(_stream_handle)
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf(t):
    assert t.dtype == torch.bfloat16, t.dtype
    return t


# --------------------------------------------------------------------------- GEMM
def gemm(layout, A, B, C, M, N, K, lda, ldb, ldc, bias=None, alpha=1.0, alpha_ncols=-1, resid=None, ldr=0,
         flags=0, batch=1, sA=0, sB=0, sC=0, sR=0, splitk=1):
    rc = lib().ifseg_gemm_bf16(c_int(layout), _ptr(A), _ptr(B), _ptr(C), c_int(M), c_int(N), c_int(K),
                               c_int(lda), c_int(ldb), c_int(ldc), _ptr(bias), c_float(alpha), c_int(alpha_ncols),
                               _ptr(resid), c_int(ldr), c_int(flags), c_int(batch), c_ll(sA), c_ll(sB), c_ll(sC),
                               c_ll(sR), c_int(splitk), _stream())
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


def linear_dx_rowdot(dy, w, out, dot, dot_out, rows_per_batch):
    """dx[M,K] = dy[M,N] @ w[N,K] and dot_out[b, h, t] = sum_c dx[b*T+t, 64h+c] * dot[b*T+t, 64h+c] (attention delta)"""
    M, N = dy.shape
    K = w.shape[1]
    rc = lib().ifseg_gemm_nn_rowdot(_ptr(_bf(dy)), _ptr(_bf(w)), _ptr(out), c_int(M), c_int(K), c_int(N), c_int(dy.stride(0)),
                                    c_int(w.stride(0)), c_int(out.stride(0)), _ptr(dot), c_int(dot.stride(0)), _ptr(dot_out),
                                    c_int(rows_per_batch), _stream())
    _check(rc, "gemm_nn_rowdot")
    return out


def ffn_ln_coef(w2, gamma, beta, b2, coef):
    """coef fp32 [2, J]: row sums of W2 [J, N] against gamma / beta (+ b2) -- see include/ifseg_hip.h.  Lists of equally
    shaped tensors (one entry per layer, <= 32) go out in ONE launch."""
    if not isinstance(w2, (list, tuple)):
        w2, gamma, beta, b2, coef = [w2], [gamma], [beta], [b2], [coef]
    L = len(w2)
    J, N = w2[0].shape
    arr = lambda ts: (c_void_p * L)(*[t.data_ptr() for t in ts])
    _check(lib().ifseg_ffn_ln_coef(arr([_bf(t) for t in w2]), c_int(w2[0].stride(0)), arr(gamma), arr(beta), arr(b2), arr(coef),
                                   c_int(L), c_int(J), c_int(N), _stream()), "ffn_ln_coef")
    return coef


def ffn_ln_rowstats(dy, t, coef, c, N):
    """c fp32 [rows, 2]: the two row means of the ffn_layernorm backward from dY [rows, J] and the saved fc2 output t"""
    rows, J = dy.shape
    _check(lib().ifseg_ffn_ln_rowstats(_ptr(_bf(dy)), c_int(dy.stride(0)), _ptr(_bf(t)), c_int(t.stride(0)), _ptr(coef), _ptr(c),
                                       c_int(rows), c_int(J), c_int(N), _stream()), "ffn_ln_rowstats")
    return c


def linear_dx_gelu_ln_bwd(dy, w, out, u, gamma, mean, rstd, c):
    """du[M, N] = LayerNorm+GELU backward of dz = dy[M, K] @ w[K, N], in the GEMM epilogue (dz is not written)"""
    M, K = dy.shape
    N = w.shape[1]
    _check(lib().ifseg_gemm_nn_gelu_ln_bwd(_ptr(_bf(dy)), _ptr(_bf(w)), _ptr(out), c_int(M), c_int(N), c_int(K),
                                           c_int(dy.stride(0)), c_int(w.stride(0)), c_int(out.stride(0)), _ptr(_bf(u)),
                                           c_int(u.stride(0)), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(c), _stream()),
           "gemm_nn_gelu_ln_bwd")
    return out


_pg_ws = {}


def ffn_ln_param_grads(w2, dw2, db2, gamma, beta, dgamma, dbeta, dy=None, u=None, mean=None, rstd=None):
    """dy [M, J], u [M, N], mean / rstd [M]: the operands of the rescue path for gains too small to divide by (csrc/ffn_ln.hip)"""
    J, N = w2.shape
    assert w2.is_contiguous() and dw2.is_contiguous()
    if J > 1024:
        # the rescue kernel stages one fc2.weight column (J values) in LDS: embed dims above 1024 (segofa_huge: 1280) run without
        # it -- a gain of exactly 0 then yields a zero gradient instead of the value from the definition
        if dy is not None and not _pg_ws.get("warned"):
            _pg_ws["warned"] = True
            import warnings
            warnings.warn("ifseg_amd: ffn_layernorm gradients run without the small-gain rescue path for embed dims above 1024 "
                          "(J = %d): a gain of exactly 0 yields a zero gradient there" % J)
        dy = None
    ws = _pg_ws.get((w2.device, N))
    if ws is None:
        ws = _pg_ws[(w2.device, N)] = torch.empty(16 * N, dtype=torch.float32, device=w2.device)
    _check(lib().ifseg_ffn_ln_param_grads(_ptr(_bf(w2)), _ptr(_bf(dw2)), _ptr(_bf(db2)), _ptr(gamma), _ptr(beta), _ptr(dgamma),
                                          _ptr(dbeta), _ptr(ws), c_int(J), c_int(N),
                                          _ptr(_bf(dy)) if dy is not None else None, c_int(dy.stride(0) if dy is not None else 0),
                                          _ptr(_bf(u)) if u is not None else None, c_int(u.stride(0) if u is not None else 0),
                                          _ptr(mean), _ptr(rstd), c_int(dy.shape[0] if dy is not None else 0), _stream()),
           "ffn_ln_param_grads")


_splitk_ws = {}


GEMM_COLSUM = 8
# workgroups a dW GEMM is split over: it runs on the side stream next to the dX chain, so filling the chip alone is not
# the goal (256 / 512 / 128 measured within 1% of each other; fewer slabs = less fp32 workspace traffic)
DW_TARGET_WGS = int(lab.get("DW_WGS", "256"))
DW_XCD_SLICES = lab.get("DW_NO_XCD_SLICES") is None
# dW GEMMs with at least this many 128x128 tiles run without split-K (bf16 dW and db written once, no slabs, no
# reduction).  Off by default: measured on the Base step, 144-tile GEMMs that walk all 8480 tokens per workgroup make the
# weight-gradient stream lag (372-382 img/s vs 388 with split-K 2); the path is kept for models with wider layers.
DW_DIRECT_TILES = int(lab.get("DW_DIRECT", "1000000"))


def linear_dw(dy, x, out, accumulate=False, bias_out=None):
    """dw[N,K] = dy[M,N]^T @ x[M,K]   (out bf16 or fp32 decided by out.dtype); with ``bias_out`` also
    db[N] = colsum(dy).  The reduction runs over the M tokens; with few output tiles it is split over
    workgroups (split-K into an fp32 workspace, then one reduction pass).  When ``bias_out`` sits right behind
    ``out`` in memory (a Linear's weight and bias in the gradient arena) the column sums ride on the same
    GEMM (one extra MFMA against an all-ones fragment) and the same reduction pass.
    Returns True if db was produced, False if the caller still has to compute it."""
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    splitk = max(1, min(16, DW_TARGET_WGS // max(1, tiles), M // 512))
    if DW_XCD_SLICES and M >= 4096:
        # k-slices pinned to XCDs (ifseg_gemm_bf16, TN): 8 slices for few-tile products, else 4 or 2; the count must divide
        # 8 and 8/splitk must divide the tile count
        for cand in (8, 4, 2):
            if tiles * cand <= max(DW_TARGET_WGS, 288) * (1 if cand == 8 else 2) and tiles % (8 // cand) == 0:
                splitk = cand
                break
    adjacent = (bias_out is not None and N % 4 == 0 and bias_out.dtype == out.dtype and bias_out.is_contiguous()
                and out.is_contiguous() and bias_out.data_ptr() == out.data_ptr() + out.numel() * out.element_size())
    if tiles >= DW_DIRECT_TILES and out.dtype == torch.bfloat16 and out.is_contiguous():
        # enough tiles for a side-stream GEMM: one pass over all M tokens per workgroup, dW (and db) written once as bf16
        # -- no fp32 slabs, no reduction launch
        gemm(GEMM_TN, _bf(dy), _bf(x), out, N, K, M, dy.stride(0), x.stride(0), K,
             flags=(GEMM_ACCUMULATE if accumulate else 0) | (GEMM_COLSUM if adjacent else 0))
        return adjacent
    if bias_out is not None and M >= 1024:
        splitk = max(splitk, 2)      # the fused column sums ride on the split-K reduction
    if splitk > 1 and out.is_contiguous():
        kchunk = (((M + splitk - 1) // splitk) + 63) // 64 * 64
        nsl = (M + kchunk - 1) // kchunk
        fuse_b = (bias_out is not None and N % 4 == 0 and bias_out.dtype == out.dtype and bias_out.is_contiguous()
                  and bias_out.data_ptr() == out.data_ptr() + out.numel() * out.element_size())
        slab = N * K + (N if fuse_b else 0)
        ws = _splitk_ws.get(dy.device)
        if ws is None or ws.numel() < nsl * slab:
            ws = torch.empty(max(nsl * slab, 16 * (3072 * 768 + 3072)), dtype=torch.float32, device=dy.device)
            _splitk_ws[dy.device] = ws
        gemm(GEMM_TN, _bf(dy), _bf(x), ws, N, K, M, dy.stride(0), x.stride(0), K,
             flags=GEMM_OUT_F32 | (GEMM_COLSUM if fuse_b else 0), splitk=splitk)
        dst = out.as_strided((slab,), (1,)) if fuse_b else out       # dW followed by db, contiguous in the arena
        reduce_parts(ws, dst, 1, nsl, slab, accumulate=accumulate)
        return fuse_b
    flags = (GEMM_OUT_F32 if out.dtype == torch.float32 else 0) | (GEMM_ACCUMULATE if accumulate else 0)
    gemm(GEMM_TN, _bf(dy), _bf(x), out, N, K, M, dy.stride(0), x.stride(0), out.stride(0), flags=flags)
    return False


class _TnProblem(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p)] + [(n, c_int) for n in ("M", "N", "K", "lda", "ldb", "colsum", "accumulate")]


GEMM_GROUP_MAX = 8
DW_GROUP_WGS = int(lab.get("DW_GROUP_WGS", "512"))     # grid cap of the grouped dW GEMM: two workgroups per CU (in-step sweep, round 3: 256: 17.49, 384: 17.46, 512: 17.23 / 17.39, 768: 17.28, uncapped: 17.36 ms)


def dw_groupable(dy, x, out, bias_out):
    """can this dW = dy^T x (+ db) ride in a grouped launch?  bf16 dW contiguous in the arena, db right behind it"""
    if out.dtype != torch.bfloat16 or not out.is_contiguous() or dy.stride(1) != 1 or x.stride(1) != 1:
        return False
    if bias_out is not None and not (bias_out.is_contiguous() and bias_out.dtype == out.dtype and
                                     bias_out.data_ptr() == out.data_ptr() + out.numel() * out.element_size()):
        return False
    return dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0


def linear_dw_group(tasks, wgs=None):
    """tasks: list of (dy [M,N], x [M,K], out [N,K] bf16, bias_out or None): every dW (+ db) in ONE launch
    (ifseg_gemm_tn_group), at most GEMM_GROUP_MAX per launch.  `wgs`: workgroup cap (default DW_GROUP_WGS; 0 = one workgroup per
    tile, for a launch with the GPU to itself)"""
    for i in range(0, len(tasks), GEMM_GROUP_MAX):
        chunk = tasks[i:i + GEMM_GROUP_MAX]
        arr = (_TnProblem * len(chunk))()
        for q, (dy, x, out, bias_out) in zip(arr, chunk):
            M, N = dy.shape
            q.A, q.B, q.C = _p(_bf(dy)), _p(_bf(x)), _p(out)
            q.M, q.N, q.K, q.lda, q.ldb = N, x.shape[1], M, dy.stride(0), x.stride(0)
            q.colsum, q.accumulate = (1 if bias_out is not None else 0), 0
        _check(lib().ifseg_gemm_tn_group(c_int(len(chunk)), arr, c_int(DW_GROUP_WGS if wgs is None else wgs), _stream()), "gemm_tn_group")


def conv2d_nhwc(x, w, shift, resid, out, B, H, W, Cin, Cout, KH, KW, stride, pad, relu):
    rc = lib().ifseg_conv2d_nhwc_bf16(_ptr(x), _ptr(w), _ptr(shift), _ptr(resid), _ptr(out), c_int(B), c_int(H),
                                      c_int(W), c_int(Cin), c_int(Cout), c_int(KH), c_int(KW), c_int(stride),
                                      c_int(pad), c_int(1 if relu else 0), _stream())
    _check(rc, "conv2d_nhwc")
    return out


# ---------------------------------------------------------------------- attention
class RelBias:
    """Per-(layer) rel-pos bias operands of the attention kernels (all device tensors)."""

    def __init__(self, P, gcode, code_bias, rel2d, rel1d, relx, grid_w=0):
        self.P, self.gcode, self.code_bias, self.grid_w = P, gcode, code_bias, grid_w
        self.rel2d, self.rel1d, self.relx = rel2d, rel1d, relx   # fp32 [H,n2d], [H,2Lt-1], [H,2]


_F32_KEEP = []


def _f32(t):
    """[H] head gains are read as fp32 by the kernels; a bf16 tensor (tests) is converted (and kept alive until the next call)"""
    if t is None or t.dtype == torch.float32:
        return t
    c = t.float()
    _F32_KEEP[:] = [c]
    return c


def attn_fwd(q, k, v, pos_q, pos_k, out, lse, B, H, T, S, rel=None, causal=False, P=None, dense_bias=None,
             gain=None):
    """q/k/v/out: [B, T|S, *] bf16 row-strided views (head h at cols h*64..); pos_q/pos_k: [T|S, H*64]."""
    return attn_fwd_gain(q, k, v, pos_q, pos_k, out, lse, B, H, T, S, rel, causal, P, dense_bias, gain)


class _AttnBwdArgs(ctypes.Structure):
    _fields_ = ([(n, c_void_p) for n in ("q", "k", "v", "pos_q", "pos_k", "out", "dout", "lse", "delta", "dq", "dk",
                                          "dv", "dpos_q_part", "dpos_k_part")]
                + [(n, c_int) for n in ("B", "H", "T", "S", "ldq", "ldk", "ldv", "ldpq", "ldpk", "ldout", "lddo",
                                        "lddq", "lddk", "lddv")]
                + [(n, c_ll) for n in ("q_bs", "k_bs", "v_bs", "out_bs", "do_bs", "dq_bs", "dk_bs", "dv_bs")]
                + [(n, c_int) for n in ("rel_mode", "P", "code_bias", "n2d", "causal", "nparts")]
                + [(n, c_void_p) for n in ("gcode", "rel2d", "rel1d", "relx", "gain", "drel2d_part", "drel1d_part",
                                           "drelx_part")]   # field order == ifseg_attn_bwd_args
                + [("dq_scale", c_float), ("dpq_scale", c_float), ("grid_w", c_int), ("phases", c_int), ("dgain_rows", c_void_p)])


def _p(t):
    return t.data_ptr() if t is not None else None


def attn_fwd_gain(q, k, v, pos_q, pos_k, out, lse, B, H, T, S, rel=None, causal=False, P=None, dense_bias=None,
                  gain=None):
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
                          _ptr(_f32(gain)), c_int(rel.grid_w if rel is not None else 0),
                          c_int(dense_bias.stride(1) if dense_bias is not None else 0), _stream())
    _check(rc, "attn_fwd")
    return out


ATTN_BWD_DELTA, ATTN_BWD_DKV, ATTN_BWD_DQ = 1, 2, 4


def attn_bwd(q, k, v, pos_q, pos_k, out, dout, lse, delta, dq, dk, dv, dpq_part, dpk_part, B, H, T, S, rel=None,
             causal=False, P=None, gain=None, dq_scale=1.0, dpq_scale=1.0, drel2d_part=None, drel1d_part=None,
             drelx_part=None, nparts=0, phases=0, dgain_rows=None):
    a = _AttnBwdArgs()
    a.dgain_rows = _p(dgain_rows)
    assert all(t is None or t.dtype == torch.bfloat16 for t in (dpq_part, dpk_part)), "abs-pos partials are bf16"
    if rel is not None:
        P = rel.P
    if P is None:
        P = S
    for name, t in (("q", q), ("k", k), ("v", v), ("pos_q", pos_q), ("pos_k", pos_k), ("out", out), ("dout", dout),
                    ("lse", lse), ("delta", delta), ("dq", dq), ("dk", dk), ("dv", dv), ("dpos_q_part", dpq_part),
                    ("dpos_k_part", dpk_part), ("gain", _f32(gain)), ("drel2d_part", drel2d_part),
                    ("drel1d_part", drel1d_part), ("drelx_part", drelx_part)):
        setattr(a, name, _p(t))
    a.B, a.H, a.T, a.S = B, H, T, S
    a.ldq, a.ldk, a.ldv = q.stride(1), k.stride(1), v.stride(1)
    a.ldpq = pos_q.stride(0) if pos_q is not None else 0
    a.ldpk = pos_k.stride(0) if pos_k is not None else 0
    a.ldout, a.lddo, a.lddq, a.lddk, a.lddv = out.stride(1), dout.stride(1), dq.stride(1), dk.stride(1), dv.stride(1)
    a.q_bs, a.k_bs, a.v_bs, a.out_bs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.do_bs, a.dq_bs, a.dk_bs, a.dv_bs = dout.stride(0), dq.stride(0), dk.stride(0), dv.stride(0)
    a.rel_mode = 1 if rel is not None else 0
    a.P, a.causal, a.nparts = P, 1 if causal else 0, nparts
    if rel is not None:
        a.code_bias, a.n2d = rel.code_bias, rel.rel2d.shape[1]
        a.gcode, a.rel2d, a.rel1d, a.relx = _p(rel.gcode), _p(rel.rel2d), _p(rel.rel1d), _p(rel.relx)
        a.grid_w = rel.grid_w
    a.dq_scale, a.dpq_scale = dq_scale, dpq_scale
    a.phases = phases
    rc = lib().ifseg_attn_bwd(ctypes.byref(a), _stream())
    _check(rc, "attn_bwd")


# ---- attention backward with the batch as a workgroup's inner dimension (csrc/attention_bi.hip)
def _pad32(n):
    return (n + 31) // 32 * 32


class DenseBias:
    """The batch-invariant attention bias of one layer as a dense bf16 operand D [H, Tp, Sp] (-inf = masked or padding),
    built from parameters only by ifseg_attn_dense_bias (fp32 arithmetic, one rounding)."""

    def __init__(self, H, T, S, device):
        self.H, self.T, self.S, self.Sp, self.Tp = H, T, S, _pad32(S), _pad32(T)
        self.D = torch.empty(H, self.Tp, self.Sp, dtype=torch.bfloat16, device=device)


def attn_dense_bias(dense, pos_q, pos_k, rel=None, causal=False, P=None):
    if rel is not None:
        P = rel.P
    if P is None:
        P = dense.S
    rc = lib().ifseg_attn_dense_bias(
        _ptr(pos_q), _ptr(pos_k), c_int(pos_q.stride(0) if pos_q is not None else 0),
        c_int(pos_k.stride(0) if pos_k is not None else 0), c_int(dense.H), c_int(dense.T), c_int(dense.S),
        c_int(1 if rel is not None else 0), c_int(P), _ptr(rel.gcode) if rel is not None else None,
        c_int(rel.code_bias if rel is not None else 0), c_int(rel.rel2d.shape[1] if rel is not None else 0),
        _ptr(rel.rel2d) if rel is not None else None, _ptr(rel.rel1d) if rel is not None else None,
        _ptr(rel.relx) if rel is not None else None, c_int(1 if causal else 0), _ptr(dense.D), c_int(dense.Sp),
        c_int(dense.Tp), _stream())
    _check(rc, "attn_dense_bias")
    return dense


class _AttnBiArgs(ctypes.Structure):
    _fields_ = ([(n, c_void_p) for n in ("q", "k", "v", "dout", "lse", "delta", "D", "gain", "dq", "dk", "dv", "dbias")]
                + [(n, c_int) for n in ("B", "H", "T", "S", "Sp", "Tp", "ldq", "ldk", "ldv", "lddo", "lddq", "lddk", "lddv")]
                + [(n, c_ll) for n in ("q_bs", "k_bs", "v_bs", "do_bs", "dq_bs", "dk_bs", "dv_bs")]
                + [("causal", c_int), ("P", c_int), ("dq_scale", c_float), ("phases", c_int), ("dgain_rows", c_void_p),
                   ("out", c_void_p), ("ldout", c_int), ("out_bs", c_ll), ("kv_len", c_void_p), ("drop_p", c_float),
                   ("drop_seed", ctypes.c_uint64), ("drop_seed_add", c_void_p)])   # == ifseg_attn_bi_args


def _attn_drop(a, drop):
    """drop = (p, seed) or None: attention dropout of the batch-inner kernels (the per-update seed word is the global seed_add)"""
    if drop is not None and drop[0] > 0:
        a.drop_p, a.drop_seed, a.drop_seed_add = float(drop[0]), drop[1] & 0xFFFFFFFFFFFFFFFF, _seed_add_ptr()


def attn_dropout_mask(B, H, T, S, p, seed, device):
    """uint8 [B, H, T, S]: the keep mask the attention kernels apply for (p, seed) and the current seed_add"""
    out = torch.empty(B, H, T, S, dtype=torch.uint8, device=device)
    _check(lib().ifseg_attn_dropout_mask(_ptr(out), c_int(B), c_int(H), c_int(T), c_int(S), c_float(p),
                                         ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), c_void_p(_seed_add_ptr()), _stream()), "attn_dropout_mask")
    return out


def _kvlen(kv_len, B):
    if kv_len is None:
        return None
    assert kv_len.dtype == torch.int32 and kv_len.is_cuda and kv_len.numel() == B and kv_len.is_contiguous()
    return kv_len


def attn_bwd_bi(q, k, v, dout, lse, delta, dense, dq, dk, dv, dbias, B, H, T, S, causal=False, P=None, gain=None,
                dq_scale=1.0, phases=0, dgain_rows=None, kv_len=None, drop=None):
    """dbias: bf16 [ceil(B/4), H, T, dense.Sp] -- zero-filled once by the caller when causal (skipped blocks are not written);
    kv_len: int32 [B] valid key counts (key padding), None = no padding"""
    a = _AttnBiArgs()
    for name, t in (("q", q), ("k", k), ("v", v), ("dout", dout), ("lse", lse), ("delta", delta), ("D", dense.D),
                    ("gain", _f32(gain)), ("dq", dq), ("dk", dk), ("dv", dv), ("dbias", dbias)):
        setattr(a, name, _p(t))
    assert dbias.dtype == torch.bfloat16 and tuple(dbias.shape) == ((B + 3) // 4, H, T, dense.Sp) and dbias.is_contiguous()
    a.B, a.H, a.T, a.S, a.Sp, a.Tp = B, H, T, S, dense.Sp, dense.Tp
    a.ldq, a.ldk, a.ldv, a.lddo = q.stride(1), k.stride(1), v.stride(1), dout.stride(1)
    a.lddq, a.lddk, a.lddv = dq.stride(1), dk.stride(1), dv.stride(1)
    a.q_bs, a.k_bs, a.v_bs, a.do_bs = q.stride(0), k.stride(0), v.stride(0), dout.stride(0)
    a.dq_bs, a.dk_bs, a.dv_bs = dq.stride(0), dk.stride(0), dv.stride(0)
    a.causal, a.P = (1 if causal else 0), (P if P is not None else S)
    a.dq_scale, a.phases = dq_scale, phases
    a.dgain_rows = _p(dgain_rows)
    a.kv_len = _p(_kvlen(kv_len, B))
    _attn_drop(a, drop)
    _check(lib().ifseg_attn_bwd_bi(ctypes.byref(a), _stream()), "attn_bwd_bi")


def dbias_nparts():
    return lib().ifseg_attn_dbias_nparts()


def attn_fwd_bi(q, k, v, dense, out, lse, B, H, T, S, causal=False, P=None, gain=None, kv_len=None, drop=None):
    """out = gain softmax(q k^T + dense.D) v, four batch elements per workgroup (csrc/attention_bi.hip)"""
    a = _AttnBiArgs()
    for name, t in (("q", q), ("k", k), ("v", v), ("lse", lse), ("D", dense.D), ("gain", _f32(gain)), ("out", out)):
        setattr(a, name, _p(t))
    a.B, a.H, a.T, a.S, a.Sp, a.Tp = B, H, T, S, dense.Sp, dense.Tp
    a.ldq, a.ldk, a.ldv, a.ldout = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    a.q_bs, a.k_bs, a.v_bs, a.out_bs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.causal, a.P = (1 if causal else 0), (P if P is not None else S)
    a.kv_len = _p(_kvlen(kv_len, B))
    _attn_drop(a, drop)
    _check(lib().ifseg_attn_fwd_bi(ctypes.byref(a), _stream()), "attn_fwd_bi")
    return out


class _AttnDbiasArgs(ctypes.Structure):
    _fields_ = [("dbias", c_void_p), ("ng", c_int), ("H", c_int), ("T", c_int), ("S", c_int), ("Sp", c_int), ("C", c_int),
                ("pos_q", c_void_p), ("pos_k", c_void_p), ("ldpq", c_int), ("ldpk", c_int),
                ("dpos_q_acc", c_void_p), ("dpos_k_acc", c_void_p), ("accumulate_pos", c_int), ("dpq_scale", c_float),
                ("P", c_int), ("grid_h", c_int), ("grid_w", c_int),
                ("drel2d", c_void_p), ("drel1d", c_void_p), ("drelx", c_void_p), ("causal", c_int)]     # == ifseg_attn_dbias_args


def attn_dbias_grads(dbias, S, pos_q=None, pos_k=None, dpq_acc=None, dpk_acc=None, accumulate_pos=False, dpq_scale=1.0,
                     P=0, grid_h=0, grid_w=0, drel2d=None, drel1d=None, drelx=None, causal=False):
    """dbias [ng, H, T, Sp] bf16 -> abs-pos operand gradients (fp32 [T, C] / [S, C]) and delta-table gradients as
    NP = dbias_nparts() partial tables per head (fp32 [H, NP, (2gh-1)(2gw-1)], [H, NP, 2Lt-1], [H, NP, 2]) in one launch"""
    if drel2d is not None:
        assert all(t.shape[1] == dbias_nparts() and t.is_contiguous() for t in (drel2d, drel1d, drelx))
    a = _AttnDbiasArgs()
    ng, H, T, Sp = dbias.shape
    a.dbias, a.ng, a.H, a.T, a.S, a.Sp = _p(dbias), ng, H, T, S, Sp
    a.C = dpq_acc.shape[1] if dpq_acc is not None else 0
    a.pos_q, a.pos_k = _p(pos_q), _p(pos_k)
    a.ldpq = pos_q.stride(0) if pos_q is not None else 0
    a.ldpk = pos_k.stride(0) if pos_k is not None else 0
    a.dpos_q_acc, a.dpos_k_acc = _p(dpq_acc), _p(dpk_acc)
    a.accumulate_pos, a.dpq_scale = (1 if accumulate_pos else 0), dpq_scale
    a.P, a.grid_h, a.grid_w = P, grid_h, grid_w
    a.drel2d, a.drel1d, a.drelx = _p(drel2d), _p(drel1d), _p(drelx)
    a.causal = 1 if causal else 0
    _check(lib().ifseg_attn_dbias_grads(ctypes.byref(a), _stream()), "attn_dbias_grads")


# ------------------------------------------------------------------------ row ops
def _map(t, rpb):
    """(bs, ld) of a [rows, C] or [B, rpb, C] view"""
    if t is None:
        return 0, 0
    if t.dim() == 3:
        return t.stride(0), t.stride(1)
    return 0, t.stride(0)


class _DropArgs(ctypes.Structure):
    _fields_ = [("p", c_float), ("seed", ctypes.c_ulonglong), ("drop_path_scale", c_void_p), ("rows_per_batch", c_int),
                ("seed_add", c_void_p)]


_seed_add = [None]


def set_seed_add(t):
    """device int64 / uint64 word added to every dropout / DropPath seed (the per-update part of the seed: the engine
    keeps it in device memory so that a captured step draws new masks on every replay); None: seeds are used as given"""
    prev = _seed_add[0]
    _seed_add[0] = t
    return prev


def _seed_add_ptr():
    t = _seed_add[0]
    return t.data_ptr() if t is not None else None


def _drop_ref(drop, rows):
    """drop = (p, seed, drop_path_scale tensor or None, rows_per_batch or None) -> byref(ifseg_drop_args) / None"""
    if drop is None:
        return None
    p, seed, dps, rpb = drop
    return ctypes.byref(_DropArgs(p, seed & 0xFFFFFFFFFFFFFFFF, dps.data_ptr() if dps is not None else None, rpb or rows,
                                  _seed_add_ptr()))


def _ln_flags(gamma, gelu=False, other=None):
    """IFSEG_LN_GELU | IFSEG_LN_PARAMS_F32 (gains / biases given as fp32 tensors: the master copy)"""
    f32 = gamma is not None and gamma.dtype == torch.float32
    if other is not None and gamma is not None:
        assert (other.dtype == torch.float32) == f32
    elif other is not None:
        f32 = other.dtype == torch.float32
    return (1 if gelu else 0) | (2 if f32 else 0)


def ln_fwd(x, gamma, beta, y, mean=None, rstd=None, resid=None, gelu=False, eps=1e-5, drop=None):
    """x, y, resid: [rows, C] or [B, rpb, C] (strided views allowed, last dim contiguous).
    drop: fused dropout/DropPath of the normalised output before the residual add (see _drop_ref)."""
    C = x.shape[-1]
    rows = x.numel() // C
    rpb = x.shape[1] if x.dim() == 3 else 0
    xb, xl = _map(x, rpb)
    yb, yl = _map(y, rpb)
    rb, rl = _map(resid, rpb)
    rc = lib().ifseg_ln_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(resid), _ptr(y), _ptr(mean), _ptr(rstd),
                            c_int(rows), c_int(C), c_float(eps), c_int(_ln_flags(gamma, gelu)), c_int(rpb), c_ll(xb),
                            c_int(xl), c_ll(yb), c_int(yl), c_ll(rb), c_int(rl), _drop_ref(drop, rpb or rows), _stream())
    _check(rc, "ln_fwd")
    return y


def ln_fwd_pair(x, gamma, beta, y, mean, rstd, gamma2, beta2, y2, mean2, rstd2, resid=None, eps=1e-5, drop=None):
    """y = [resid +] drop(LN(x)), y2 = LN2(y): the post-LN of a block and the pre-LN of the next in one launch"""
    C = x.shape[-1]
    rows = x.numel() // C
    rpb = x.shape[1] if x.dim() == 3 else 0
    xb, xl = _map(x, rpb)
    yb, yl = _map(y, rpb)
    rb, rl = _map(resid, rpb)
    y2b, y2l = _map(y2, rpb)
    rc = lib().ifseg_ln_fwd_pair(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(resid), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma2),
                                 _ptr(beta2), _ptr(y2), _ptr(mean2), _ptr(rstd2), c_int(rows), c_int(C), c_float(eps),
                                 c_int(_ln_flags(gamma, False, gamma2)), c_int(rpb), c_ll(xb), c_int(xl), c_ll(yb), c_int(yl), c_ll(rb), c_int(rl), c_ll(y2b),
                                 c_int(y2l), _drop_ref(drop, rpb or rows), _stream())
    _check(rc, "ln_fwd_pair")
    return y, y2


LN_BWD_BLOCKS = int(lab.get("LN_BWD_BLOCKS", "768"))   # three resident blocks per CU (146-162 VGPRs): best of a 256..2048 sweep on MI355X


def ln_bwd_drop(dy, x, gamma, mean, rstd, dx, dgamma_part, dbeta_part, dx2, dx_add=None, drop2=None):
    """dx = [dx_add +] LN'(x; gamma)(dy) and dx2 = drop2(dx): the pre-LN backward that closes a block of the backward and the
    fc2-dropout adjoint that opens the next one in one launch"""
    C = x.shape[-1]
    rows = x.numel() // C
    rpb = x.shape[1] if x.dim() == 3 else 0
    db_, dl = _map(dy, rpb)
    xb, xl = _map(x, rpb)
    ob, ol = _map(dx, rpb)
    ab, al = _map(dx_add, rpb)
    o2b, o2l = _map(dx2, rpb)
    rc = lib().ifseg_ln_bwd_drop(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx_add), _ptr(dx),
                                 _ptr(dgamma_part), _ptr(dbeta_part), _ptr(dx2), c_int(LN_BWD_BLOCKS), c_int(rows), c_int(C),
                                 c_int(_ln_flags(gamma)), c_int(rpb), c_ll(db_), c_int(dl), c_ll(xb), c_int(xl), c_ll(ob), c_int(ol), c_ll(ab),
                                 c_int(al), c_ll(o2b), c_int(o2l), _drop_ref(drop2, rpb or rows), _stream())
    _check(rc, "ln_bwd_drop")
    return dx, dx2


def ln_bwd(dy, x, gamma, mean, rstd, dx, dgamma_part, dbeta_part, dx_add=None, gelu=False, drop=None):
    C = x.shape[-1]
    rows = x.numel() // C
    rpb = x.shape[1] if x.dim() == 3 else 0
    db_, dl = _map(dy, rpb)
    xb, xl = _map(x, rpb)
    ob, ol = _map(dx, rpb)
    ab, al = _map(dx_add, rpb)
    rc = lib().ifseg_ln_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx_add), _ptr(dx),
                            _ptr(dgamma_part), _ptr(dbeta_part), c_int(LN_BWD_BLOCKS), c_int(rows), c_int(C),
                            c_int(_ln_flags(gamma, gelu)), c_int(rpb), c_ll(db_), c_int(dl), c_ll(xb), c_int(xl), c_ll(ob),
                            c_int(ol), c_ll(ab), c_int(al), _drop_ref(drop, rpb or rows), _stream())
    _check(rc, "ln_bwd")
    return dx


CHECK_ANY_EQ, CHECK_SUFFIX, CHECK_ANY_ZERO_BYTE, CHECK_FLAG = 0, 1, 2, 3


def check_inputs(x, mode, verdict_pinned, slot, value=0, row_len=1):
    """one-launch input validation (ifseg_check_inputs): verdict_pinned[slot] <- 0 / 1 in stream order"""
    n = 1 if mode == CHECK_FLAG else x.numel()
    _check(lib().ifseg_check_inputs(_ptr(x), c_ll(n), c_int(mode), c_ll(value), c_ll(row_len),
                                    c_void_p(verdict_pinned.data_ptr() + slot), _stream()), "check_inputs")


def reduce_parts(inp, out, outer, parts, n, accumulate=False, scale=1.0):
    rc = lib().ifseg_reduce_parts(_ptr(inp), _ptr(out), c_int(outer), c_int(parts), c_ll(n),
                                  c_int(1 if accumulate else 0), c_int(1 if out.dtype == torch.bfloat16 else 0),
                                  c_float(scale), _stream())
    _check(rc, "reduce_parts")
    return out


class _ReduceTask(ctypes.Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("outer", c_int), ("parts", c_int), ("n", c_ll),
                ("accumulate", c_int), ("out_bf16", c_int), ("scale", c_float)]


def reduce_parts_multi(tasks):
    """tasks: list of (inp, out, outer, parts, n, accumulate) -- ifseg_reduce_parts for each, 16 per launch"""
    for i in range(0, len(tasks), 16):
        chunk = tasks[i:i + 16]
        arr = (_ReduceTask * len(chunk))()
        for q, (inp, out, outer, parts, n, accumulate) in zip(arr, chunk):
            q.inp, q.out, q.outer, q.parts, q.n = _p(inp), _p(out), outer, parts, n
            q.accumulate, q.out_bf16, q.scale = (1 if accumulate else 0), (1 if out.dtype == torch.bfloat16 else 0), 1.0
        _check(lib().ifseg_reduce_parts_multi(c_int(len(chunk)), arr, _stream()), "reduce_parts_multi")


COLSUM_BLOCKS = 256


def colsum(x, part):
    """x [M,N] (or [B,rpb,N]) bf16 -> part [COLSUM_BLOCKS, N] fp32 partial column sums"""
    N = x.shape[-1]
    M = x.numel() // N
    rpb = x.shape[1] if x.dim() == 3 else 0
    xb, xl = _map(x, rpb)
    rc = lib().ifseg_colsum_bf16(_ptr(x), _ptr(part), c_int(COLSUM_BLOCKS), c_int(M), c_int(N), c_int(rpb), c_ll(xb),
                                 c_int(xl), _stream())
    _check(rc, "colsum")
    return part


def col_mean(x, ws=None, every=1):
    """fp32 [C] column means of x [rows, C] bf16 (two launches: partial column sums, their sum); ws: a
    [COLSUM_BLOCKS + 1, C] fp32 workspace whose last row receives the result; every > 1: the mean over every `every`-th row"""
    C = x.shape[-1]
    if ws is None:
        ws = torch.empty(COLSUM_BLOCKS + 1, C, dtype=torch.float32, device=x.device)
    rows = x.numel() // C
    if every > 1 and x.dim() == 2 and x.is_contiguous() and rows >= 64 * every:
        x = x[: rows // every * every].view(rows // every, every * C)[:, :C]       # every `every`-th row (row stride every * C)
        rows = rows // every
    colsum(x, ws[:COLSUM_BLOCKS])
    reduce_parts(ws[:COLSUM_BLOCKS], ws[COLSUM_BLOCKS], 1, COLSUM_BLOCKS, C, scale=1.0 / rows)
    return ws[COLSUM_BLOCKS]


def kproj_common_mode(gw, db, xmean):
    """gw [N, C] -= db [N] (x) xmean [C]: the key-projection weight gradient without the product of dK's spurious column sum
    and the token-common component of its input (csrc/rowops.hip: ifseg_kproj_common_mode; xmean = col_mean(x))"""
    N, C = gw.shape
    assert gw.is_contiguous() and gw.dtype == torch.bfloat16 and db.dtype == torch.bfloat16 and db.is_contiguous()
    assert xmean.dtype == torch.float32 and xmean.numel() == C
    _check(lib().ifseg_kproj_common_mode(_ptr(gw), _ptr(db), _ptr(xmean), c_int(N), c_int(C), _stream()), "kproj_common_mode")


def embed_bag_mean(table, ids, ends, add, out):
    """table [V,C] bf16; ids int64 [B,maxlen]; ends int64 [B,P] (per-sample cumulative bag ends); add [C] bf16 or None;
    out [B*P, C] or [B, P, C] bf16 view"""
    B, maxlen = ids.shape
    P = ends.numel() // B
    C = table.shape[1]
    rpb = out.shape[1] if out.dim() == 3 else 0
    ob, ol = _map(out, rpb)
    assert ids.dtype == torch.int64 and ends.dtype == torch.int64 and ids.is_contiguous() and ends.is_contiguous()
    rc = lib().ifseg_embed_bag_mean(_ptr(table), _ptr(ids), _ptr(ends), _ptr(add), _ptr(out), c_int(B), c_int(P), c_int(C),
                                    c_int(maxlen), c_int(rpb), c_ll(ob), c_int(ol), _stream())
    _check(rc, "embed_bag_mean")
    return out


def embed_rows(table, ids, add, out):
    C = table.shape[1]
    n = ids.numel()
    rpb = out.shape[1] if out.dim() == 3 else 0
    ob, ol = _map(out, rpb)
    rc = lib().ifseg_embed_rows(_ptr(table), _ptr(ids), _ptr(add), _ptr(out), c_int(n), c_int(C), c_int(rpb),
                                c_ll(ob), c_int(ol), _stream())
    _check(rc, "embed_rows")
    return out


def rows_segment_sum(src, src_row, weight, sorted_ids, table_grad, skip_id=-1):
    """table_grad[sorted_ids[j]] += weight[j] * src[src_row[j]] over runs of equal (sorted) ids; see csrc/rowops.hip"""
    C = src.shape[-1]
    assert src.is_contiguous() and table_grad.is_contiguous() and table_grad.shape[1] == C
    assert src_row.dtype == torch.int64 and sorted_ids.dtype == torch.int64 and (weight is None or weight.dtype == torch.float32)
    _check(lib().ifseg_rows_segment_sum(_ptr(src), _ptr(src_row), _ptr(weight), _ptr(sorted_ids), _ptr(table_grad),
                                        c_ll(sorted_ids.numel()), c_int(C), c_ll(table_grad.shape[0]), c_ll(skip_id), _stream()),
           "rows_segment_sum")


def sync_master(master, p16):
    _check(lib().ifseg_sync_master(_ptr(master), _ptr(p16), c_ll(master.numel()), _stream()), "sync_master")


def cast_f32_bf16(x, out, scale=1.0):
    _check(lib().ifseg_cast_f32_bf16(_ptr(x), _ptr(out), c_ll(x.numel()), c_float(scale), _stream()), "cast")
    return out


def add_bf16(a, b, out):
    _check(lib().ifseg_add_bf16(_ptr(a), _ptr(b), _ptr(out), c_ll(a.numel()), _stream()), "add")
    return out


def nchw_to_nhwc(x, out, Cpad):
    B, C, H, W = x.shape
    _check(lib().ifseg_nchw_to_nhwc_bf16(_ptr(x), c_int(1 if x.dtype == torch.float32 else 0), _ptr(out), c_int(B),
                                         c_int(C), c_int(H), c_int(W), c_int(Cpad), _stream()), "nchw_to_nhwc")
    return out


def stem_weights_mfma(w):
    """fp32 [7][7][3][64] (BN scale folded) -> bf16 [3][64][224] for ifseg_stem_conv7x7_mfma: k = ky*32 + kx*4 + ci, three bf16
    terms whose sum is the fp32 weight"""
    wt = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=w.device)
    wt[:, :, :7, :3] = w.permute(3, 0, 1, 2)
    wt = wt.reshape(64, 224)
    t0 = wt.to(torch.bfloat16)
    t1 = (wt - t0.float()).to(torch.bfloat16)
    t2 = (wt - t0.float() - t1.float()).to(torch.bfloat16)
    return torch.stack([t0, t1, t2]).contiguous()


def stem_conv(x4, w, shift, out, B, H, W):
    """w: fp32 [7][7][3][64] (direct kernel) or bf16 [3][64][224] from stem_weights_mfma (matrix cores)"""
    if w.dtype == torch.bfloat16:
        assert tuple(w.shape) == (3, 64, 224) and w.is_contiguous()
        _check(lib().ifseg_stem_conv7x7_mfma(_ptr(x4), _ptr(w), _ptr(shift), _ptr(out), c_int(B), c_int(H), c_int(W),
                                             _stream()), "stem_conv_mfma")
        return out
    _check(lib().ifseg_stem_conv7x7(_ptr(x4), _ptr(w), _ptr(shift), _ptr(out), c_int(B), c_int(H), c_int(W),
                                    _stream()), "stem_conv")
    return out


def maxpool(x, out, B, H, W, C):
    _check(lib().ifseg_maxpool3x3s2(_ptr(x), _ptr(out), c_int(B), c_int(H), c_int(W), c_int(C), _stream()), "maxpool")
    return out


def grad_sumsq(g, workspace, out):
    _check(lib().ifseg_grad_sumsq_bf16(_ptr(g), c_ll(g.numel()), _ptr(workspace), _ptr(out), _stream()), "sumsq")
    return out


def droppath_scale(out, keep, seed):
    """out fp32 [n, B] = Bernoulli(keep[i]) / keep[i]"""
    n, B = out.shape
    _check(lib().ifseg_droppath_scale(_ptr(out), _ptr(keep), c_int(n), c_int(B), ctypes.c_ulonglong(seed & 0xFFFFFFFFFFFFFFFF),
                                      c_void_p(_seed_add_ptr()), _stream()), "droppath_scale")


def adam_step(p32, g, m, v, p16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, max_norm=0.0, sumsq=None, overflow=None,
              hyper=None):
    _check(lib().ifseg_adam_step(_ptr(p32), _ptr(g), _ptr(m), _ptr(v), _ptr(p16), c_ll(p32.numel()), c_float(lr),
                                 c_float(beta1), c_float(beta2), c_float(eps), c_float(wd), c_int(step),
                                 c_float(grad_scale), c_float(max_norm), _ptr(sumsq), _ptr(overflow), _ptr(hyper), _stream()), "adam")


def rel_gather(table, idx, out):
    n, H = idx.numel(), table.shape[1]
    _check(lib().ifseg_rel_gather(_ptr(table), _ptr(idx), _ptr(out), c_int(n), c_int(H), _stream()), "rel_gather")
    return out


def resized_rel_bias(out, table2d, rel1d, relx, h, w, oh, ow, Lt, causal=False):
    """out fp32 [H, T, ld >= T] (T = h*w + Lt, rows `out.stride(1)` apart): the reference's doubly bilinear-resized rel-pos
    bias of a (h, w) grid from the delta table of the trained (oh, ow) grid; tail blocks from rel1d / relx (None: zero)"""
    H = out.shape[0]
    assert out.stride(2) == 1 and out.stride(0) == out.shape[1] * out.stride(1)
    _check(lib().ifseg_resized_rel_bias(_ptr(out), _ptr(table2d), _ptr(rel1d), _ptr(relx), c_int(H), c_int(h), c_int(w),
                                        c_int(oh), c_int(ow), c_int(Lt), c_int(1 if causal else 0), c_int(out.stride(1)),
                                        _stream()), "resized_rel_bias")
    return out


def resize_rows_bilinear(src, dst, h, w, oh, ow, src_stride, src_off):
    """dst bf16 [h*w, C] = bilinear resize of the oh x ow grid of rows src[y * src_stride + x + src_off]"""
    C = dst.shape[1]
    _check(lib().ifseg_resize_rows_bilinear(_ptr(src), _ptr(dst), c_int(C), c_int(h), c_int(w), c_int(oh), c_int(ow),
                                            c_int(src_stride), c_int(src_off), c_int(src.stride(0)), _stream()),
           "resize_rows_bilinear")
    return dst


def rel_gather_multi(tables, idx, out):
    """tables: list of L <= 16 bf16 [N, H] tensors of one shape; out fp32 [L, H, n]"""
    L, n, H = len(tables), idx.numel(), tables[0].shape[1]
    arr = (c_void_p * L)(*[t.data_ptr() for t in tables])
    _check(lib().ifseg_rel_gather_multi(arr, c_int(L), _ptr(idx), _ptr(out), c_int(n), c_int(H), _stream()), "rel_gather_multi")
    return out


class _AttnReduceArgs(ctypes.Structure):
    _fields_ = [("B", c_int), ("H", c_int), ("T", c_int), ("S", c_int), ("C", c_int), ("nparts", c_int),
                ("accumulate_pos", c_int),
                ("dpos_q_part", c_void_p), ("dpos_k_part", c_void_p), ("dpos_q_acc", c_void_p), ("dpos_k_acc", c_void_p),
                ("delta", c_void_p), ("gain", c_void_p), ("dgain", c_void_p), ("ntab", c_int),
                ("tab_part", c_void_p * 3), ("tab_idx", c_void_p * 3), ("tab_acc", c_void_p * 3), ("tab_n", c_int * 3),
                ("tab_nbucket", c_int * 3)]


def attn_bwd_reduce(B, H, T, S, C, dpq_part, dpk_part, dpq_acc, dpk_acc, accumulate_pos, delta, gain, dgain, nparts, tables):
    """every reduction behind attn_bwd's partial outputs in one launch (ifseg_attn_bwd_reduce); tables: list of
    (part [H,nparts,n] fp32, idx [n] int32, acc [n_bucket,H] fp32)"""
    assert all(t is None or t.dtype == torch.bfloat16 for t in (dpq_part, dpk_part)), "abs-pos partials are bf16"
    a = _AttnReduceArgs()
    a.B, a.H, a.T, a.S, a.C, a.nparts, a.accumulate_pos = B, H, T, S, C, nparts, 1 if accumulate_pos else 0
    a.dpos_q_part, a.dpos_k_part, a.dpos_q_acc, a.dpos_k_acc = _p(dpq_part), _p(dpk_part), _p(dpq_acc), _p(dpk_acc)
    a.delta, a.gain, a.dgain = _p(delta), _p(_f32(gain)), _p(dgain)
    a.ntab = len(tables)
    for i, (part, idx, acc) in enumerate(tables):
        assert part.dtype == torch.float32 and acc.dtype == torch.float32 and idx.dtype == torch.int32 and acc.shape[1] == H
        a.tab_part[i], a.tab_idx[i], a.tab_acc[i], a.tab_n[i], a.tab_nbucket[i] = _p(part), _p(idx), _p(acc), idx.numel(), acc.shape[0]
    _check(lib().ifseg_attn_bwd_reduce(ctypes.byref(a), _stream()), "attn_bwd_reduce")


def rel_scatter_add(d, idx, acc):
    n, H = idx.numel(), acc.shape[1]
    _check(lib().ifseg_rel_scatter_add(_ptr(d), _ptr(idx), _ptr(acc), c_int(n), c_int(H), _stream()), "rel_scatter")
    return acc


PROF_KINDS = ("gemm_nt", "gemm_nn", "gemm_tn", "conv", "attn_fwd", "attn_bwd_dkv", "attn_bwd_dq", "ln_fwd", "ln_bwd")


def prof_enable(mask, stride=1):
    lib().ifseg_prof_stride(c_int(stride))
    lib().ifseg_prof_enable(ctypes.c_uint(mask))


def prof_reset():
    lib().ifseg_prof_reset()


def prof_read(kind):
    ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    _check(lib().ifseg_prof_read(c_int(kind), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n)), "prof_read")
    return {"kind": PROF_KINDS[kind], "ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value}


def seg_loss(logits_pad, target, hp, wp, H, W, nseg, seg_id_offset, tile_partial, stats_part, stats, dlogits_pad,
             loss_out, pad_id=1, eos_id=2, bad_label=None, label_smoothing=0.0):
    """fused upsample + CE + grad + histograms; logits_pad / dlogits_pad: bf16 [B, P+1, ldl]"""
    B = logits_pad.shape[0]
    ldl = logits_pad.stride(1)
    _check(lib().ifseg_seg_loss_tiles(_ptr(logits_pad), c_int(ldl), c_ll(logits_pad.stride(0)), _ptr(target),
                                      c_ll(target.stride(0)), c_int(B), c_int(hp), c_int(wp), c_int(H), c_int(W),
                                      c_int(nseg), c_ll(seg_id_offset), c_ll(pad_id), c_ll(eos_id),
                                      _ptr(tile_partial), _ptr(stats_part), _ptr(bad_label), c_float(label_smoothing), _stream()), "seg_loss_tiles")
    reduce_parts(stats_part, stats, 1, B * hp * wp, 2 + 3 * nseg)
    _check(lib().ifseg_seg_loss_gather(_ptr(tile_partial), _ptr(stats), _ptr(dlogits_pad), c_int(dlogits_pad.stride(1)),
                                       c_ll(dlogits_pad.stride(0)), c_int(B), c_int(hp), c_int(wp), c_int(nseg),
                                       _ptr(loss_out), _stream()), "seg_loss_gather")
    return loss_out


# ---------------------------------------------------------------------- eval post-processing
def rows_to_f32(logits_pad, nseg, rows_per_batch, softmax=False, temperature=1.0):
    """logits_pad bf16 [B, T, ld] -> fp32 [B, rows_per_batch, nseg] (first rows of every sample), optionally softmaxed"""
    B = logits_pad.shape[0]
    out = torch.empty(B, rows_per_batch, nseg, dtype=torch.float32, device=logits_pad.device)
    _check(lib().ifseg_softmax_rows(_ptr(logits_pad), c_ll(logits_pad.stride(0)), c_int(rows_per_batch),
                                    c_int(logits_pad.stride(1)), _ptr(out), c_int(B * rows_per_batch), c_int(nseg),
                                    c_float(1.0 / temperature), c_int(1 if softmax else 0), _stream()), "softmax_rows")
    return out


def neighbour_smoothing(logits_pad, nseg, feat, iters, topk, temperature=1.0):
    """seg_criterion.py:197-213 on the device: logits_pad bf16 [B, P+1, ld], feat bf16 [B, P, D] (trunk features)
    -> smoothed class probabilities fp32 [B, P, nseg]"""
    B, P, D = feat.shape
    dev = feat.device
    fn = torch.empty(B * P, D, dtype=torch.bfloat16, device=dev)
    _check(lib().ifseg_l2norm_rows_bf16(_ptr(feat.contiguous()), _ptr(fn), c_int(B * P), c_int(D), _stream()), "l2norm_rows")
    sim = torch.empty(B, P, P, dtype=torch.float32, device=dev)
    gemm(GEMM_NT, fn, fn, sim, P, P, D, D, D, P, flags=GEMM_OUT_F32, batch=B, sA=P * D, sB=P * D, sC=P * P)
    idx = torch.empty(B * P, topk, dtype=torch.int32, device=dev)
    _check(lib().ifseg_topk_rows_f32(_ptr(sim), _ptr(idx), c_int(B * P), c_int(P), c_int(topk), _stream()), "topk_rows")
    prob = rows_to_f32(logits_pad, nseg, P, softmax=True, temperature=temperature)
    tmp = torch.empty_like(prob)
    for _ in range(iters):
        _check(lib().ifseg_gather_mean(_ptr(prob), _ptr(idx), _ptr(tmp), c_int(B), c_int(P), c_int(nseg), c_int(topk),
                                       _stream()), "gather_mean")
        prob, tmp = tmp, prob
    return prob


def seg_eval(scores, hp, wp, target, h, w, seg_id_offset):
    """scores fp32 [hp*wp, n] (one image), target int64 [h*w] -> (display CE loss tensor, int64 hist [3, n] =
    intersect / predicted / label pixel counts) at the original h x w resolution (seg_criterion.py:289-347)"""
    n = scores.shape[-1]
    dev = scores.device
    nblk = (h * w + 255) // 256
    hist = torch.zeros(3, n, dtype=torch.int64, device=dev)
    part = torch.empty(nblk, 2, dtype=torch.float32, device=dev)
    _check(lib().ifseg_seg_eval(_ptr(scores.contiguous()), c_int(hp), c_int(wp), c_int(n), _ptr(target.contiguous()), c_int(h),
                                c_int(w), c_ll(seg_id_offset), _ptr(hist), _ptr(part), c_int(nblk), _stream()), "seg_eval")
    tot = torch.empty(2, dtype=torch.float32, device=dev)
    reduce_parts(part, tot, 1, nblk, 2)
    return tot[0] / tot[1], hist


def dropout(x, resid, out, p, seed, drop_path_scale=None, rows_per_batch=None):
    """x / resid / out: [rows, C] or [B, rpb, C] bf16 views (last dim contiguous)"""
    C = x.shape[-1]
    rows = x.numel() // C
    rpb = x.shape[1] if x.dim() == 3 else 0
    xb, xl = _map(x, rpb)
    rb, rl = _map(resid, rpb)
    ob, ol = _map(out, rpb)
    rc = lib().ifseg_dropout(_ptr(x), _ptr(resid), _ptr(out), c_ll(rows), c_int(C), c_float(p),
                             ctypes.c_ulonglong(seed & 0xFFFFFFFFFFFFFFFF), _ptr(drop_path_scale),
                             c_int(rows_per_batch or (rpb if rpb else rows)), c_int(rpb), c_ll(xb), c_int(xl), c_ll(rb),
                             c_int(rl), c_ll(ob), c_int(ol), c_void_p(_seed_add_ptr()), _stream())
    _check(rc, "dropout")
    return out


def dropout_fill(x, out, p, seed, fill=-30.0):
    """out = keep ? x : fill (contiguous bf16, no rescaling; the activation dropout on the FFN pre-activation, see rowops.hip)"""
    assert x.is_contiguous() and out.is_contiguous() and x.dtype == torch.bfloat16 and out.shape == x.shape
    _check(lib().ifseg_dropout_fill(_ptr(x), _ptr(out), c_ll(x.numel()), c_float(p), ctypes.c_ulonglong(seed & 0xFFFFFFFFFFFFFFFF),
                                    c_void_p(_seed_add_ptr()), c_float(fill), _stream()), "dropout_fill")
    return out


# ---------------------------------------------------------------------- dense CRF (crf.py:19-37)
def crf_bilateral(feat4, gx, gy, qn, out, H, W):
    """out fp32 [Cp, N] = exact bilateral filter of qn bf16 [Cp, ldq]"""
    _check(lib().ifseg_crf_bilateral(_ptr(feat4), _ptr(gx), _ptr(gy), _ptr(qn), c_int(qn.stride(0)), _ptr(out),
                                     c_int(qn.shape[0]), c_int(H), c_int(W), _stream()), "crf_bilateral")


def crf_spatial(qn, g, R, out, H, W):
    _check(lib().ifseg_crf_spatial(_ptr(qn), _ptr(g), c_int(R), _ptr(out), c_int(qn.shape[0]), c_int(H), c_int(W), _stream()), "crf_spatial")


def crf_update(prob, mpos, mbi, npos, nbi, wpos, wbi, Q, qpos, qbi):
    C, N = prob.shape
    _check(lib().ifseg_crf_update(_ptr(prob), _ptr(mpos), _ptr(mbi), _ptr(npos), _ptr(nbi), c_float(wpos), c_float(wbi),
                                  _ptr(Q), _ptr(qpos), _ptr(qbi), c_int(qbi.stride(0) if qbi is not None else 0), c_int(C),
                                  c_int(N), _stream()), "crf_update")


def crf_norm(k1, n):
    _check(lib().ifseg_crf_norm(_ptr(k1), _ptr(n), c_int(n.numel()), _stream()), "crf_norm")
