"""torch.library registration of the hot-path kernels: `torch.ops.ifseg.*`.

BASELINE.json north_star asks for "Python host code calling hand-written HIP kernels for CDNA4 through PyTorch-ROCm custom
ops"; SURVEY 8b names torch.library.  The engine (models/segofa/engine.py) keeps calling the C ABI directly through
`ifseg_amd.hip` -- one ctypes call per launch, no dispatcher hop on a ~570-launch step -- and this module puts the same
kernels IN FRONT OF the dispatcher as functional ops with autograd and fake (meta) implementations, for callers that
compose them with other PyTorch code (`torch.compile`, `torch.autograd.gradcheck`-style tests, export):

  torch.ops.ifseg.linear(x, w, bias)                       F.linear                      unify_multihead_attention.py:327-346,513
  torch.ops.ifseg.layer_norm(x, gamma, beta, eps, gelu)    LayerNorm (+ GELU in front)   unify_transformer_layer.py:256-292
  torch.ops.ifseg.bias_attention(q, k, v, pos_q, pos_k, gain, gcode, rel2d, rel1d, relx, P, code_bias, grid_w, causal)
                                                           position-biased attention     unify_multihead_attention.py:459-512,
                                                                                         encoder_module.py:757-809
Each op has its backward registered as further ops (`linear_bwd`, `layer_norm_bwd`, `bias_attention_bwd`), so the backward
is dispatcher-visible too.  No CPU implementation is registered: on a CPU tensor the dispatcher raises.
"""
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import hip

BF = torch.bfloat16


def _stream_scope(t):
    """the kernels run on PyTorch's current stream of the tensor's device"""
    return hip.set_stream(torch.cuda.current_stream(t.device).cuda_stream)


# ----------------------------------------------------------------------------------------------- linear
@custom_op("ifseg::linear", mutates_args=(), device_types="cuda")
def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    prev = _stream_scope(x)
    try:
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        out = hip.linear_fwd(x2, w.contiguous(), bias)
        return out.view(*x.shape[:-1], w.shape[0])
    finally:
        hip.set_stream(prev)


@linear.register_fake
def _(x, w, bias):
    return x.new_empty(*x.shape[:-1], w.shape[0])


@custom_op("ifseg::linear_bwd", mutates_args=(), device_types="cuda")
def linear_bwd(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dx, dw, db): dx = g w, dw = g^T x, db = column sums of g (one GEMM carries dw and db)"""
    prev = _stream_scope(g)
    try:
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        N, K = w.shape
        dx = hip.linear_dx(g2, w.contiguous())
        buf = torch.empty(N * K + N, dtype=BF, device=g.device)          # db right behind dw: rides on the dW GEMM
        dw, db = buf[: N * K].view(N, K), buf[N * K:]
        if not hip.linear_dw(g2, x2, dw, bias_out=db):
            db.copy_(g2.float().sum(0))
        return dx.view(x.shape), dw, db.clone()           # (returns of a custom op must not share storage)
    finally:
        hip.set_stream(prev)


@linear_bwd.register_fake
def _(g, x, w):
    return x.new_empty(x.shape), w.new_empty(w.shape), w.new_empty(w.shape[0])


def _linear_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, g):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.ifseg.linear_bwd(g, x, w)
    return dx, dw, (db if ctx.has_bias else None)


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ----------------------------------------------------------------------------------------------- layer norm
@custom_op("ifseg::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, gelu: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (y, mean, rstd); gelu: y = LN(GELU(x)) (the FFN's ffn_layernorm(gelu(fc1)), GELU evaluated in fp32)"""
    prev = _stream_scope(x)
    try:
        C = x.shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        y = torch.empty_like(x2)
        mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        hip.ln_fwd(x2, gamma, beta, y, mean, rstd, gelu=gelu, eps=eps)
        return y.view(x.shape), mean, rstd
    finally:
        hip.set_stream(prev)


@layer_norm.register_fake
def _(x, gamma, beta, eps, gelu):
    rows = x.numel() // x.shape[-1]
    return x.new_empty(x.shape), x.new_empty(rows, dtype=torch.float32), x.new_empty(rows, dtype=torch.float32)


@custom_op("ifseg::layer_norm_bwd", mutates_args=(), device_types="cuda")
def layer_norm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                   gelu: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    prev = _stream_scope(x)
    try:
        C = x.shape[-1]
        x2, dy2 = x.reshape(-1, C).contiguous(), dy.reshape(-1, C).contiguous()
        dx = torch.empty_like(x2)
        part = torch.empty(2, hip.LN_BWD_BLOCKS, C, dtype=torch.float32, device=x.device)
        hip.ln_bwd(dy2, x2, gamma, mean, rstd, dx, part[0], part[1], gelu=gelu)
        dgb = torch.empty(2, C, dtype=gamma.dtype if gamma.dtype in (BF, torch.float32) else torch.float32, device=x.device)
        hip.reduce_parts(part, dgb, 2, hip.LN_BWD_BLOCKS, C)
        return dx.view(x.shape), dgb[0].clone(), dgb[1].clone()
    finally:
        hip.set_stream(prev)


@layer_norm_bwd.register_fake
def _(dy, x, gamma, mean, rstd, gelu):
    return x.new_empty(x.shape), gamma.new_empty(gamma.shape), gamma.new_empty(gamma.shape)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps, gelu = inputs
    ctx.save_for_backward(x, gamma, output[1], output[2])
    ctx.gelu = gelu


def _ln_backward(ctx, gy, gmean, grstd):
    x, gamma, mean, rstd = ctx.saved_tensors
    dx, dg, db = torch.ops.ifseg.layer_norm_bwd(gy.contiguous(), x, gamma, mean, rstd, ctx.gelu)
    return dx, dg, db, None, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ----------------------------------------------------------------------------------------------- attention
def _rel(P, gcode, rel2d, rel1d, relx, code_bias, grid_w):
    return None if gcode is None else hip.RelBias(P, gcode, code_bias, rel2d, rel1d, relx, grid_w=grid_w)


@custom_op("ifseg::bias_attention", mutates_args=(), device_types="cuda")
def bias_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos_q: torch.Tensor, pos_k: torch.Tensor,
                   gain: torch.Tensor, gcode: Optional[torch.Tensor], rel2d: Optional[torch.Tensor],
                   rel1d: Optional[torch.Tensor], relx: Optional[torch.Tensor], P: int, code_bias: int, grid_w: int,
                   causal: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """q [B,T,H*64] (already scaled), k / v [B,S,H*64], pos_q [T,H*64] (scaled), pos_k [S,H*64], gain fp32 [H];
    gcode int32 [P] + fp32 delta tables rel2d [H,n2d] / rel1d [H,2Lt-1] / relx [H,2] (or all None: no relative bias).
    -> (out [B,T,H*64] = gain_h * softmax(q k^T + pos_q pos_k^T + rel) v, lse [B,H,T] in log2 units)"""
    prev = _stream_scope(q)
    try:
        B, T, C = q.shape
        S, H = k.shape[1], C // 64
        out = torch.empty_like(q)
        lse = torch.empty(B, H, T, dtype=torch.float32, device=q.device)
        hip.attn_fwd(q, k, v, pos_q, pos_k, out, lse, B, H, T, S, rel=_rel(P, gcode, rel2d, rel1d, relx, code_bias, grid_w),
                     causal=causal, P=P, gain=gain)
        return out, lse
    finally:
        hip.set_stream(prev)


@bias_attention.register_fake
def _(q, k, v, pos_q, pos_k, gain, gcode, rel2d, rel1d, relx, P, code_bias, grid_w, causal):
    B, T, C = q.shape
    return q.new_empty(q.shape), q.new_empty(B, C // 64, T, dtype=torch.float32)


@custom_op("ifseg::bias_attention_bwd", mutates_args=(), device_types="cuda")
def bias_attention_bwd(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos_q: torch.Tensor,
                       pos_k: torch.Tensor, gain: torch.Tensor, out: torch.Tensor, lse: torch.Tensor,
                       gcode: Optional[torch.Tensor], rel2d: Optional[torch.Tensor], rel1d: Optional[torch.Tensor],
                       relx: Optional[torch.Tensor], P: int, code_bias: int, grid_w: int, causal: bool
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                                  torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dq, dk, dv, dpos_q, dpos_k, dgain, drel2d, drel1d, drelx); the three table gradients are empty tensors when the
    attention has no relative bias"""
    prev = _stream_scope(q)
    try:
        B, T, C = q.shape
        S, H = k.shape[1], C // 64
        dev = q.device
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty(B, H, T, dtype=torch.float32, device=dev)
        dpq, dpk = torch.empty(B, T, C, dtype=BF, device=dev), torch.empty(B, S, C, dtype=BF, device=dev)
        rel = _rel(P, gcode, rel2d, rel1d, relx, code_bias, grid_w)
        nparts = B * ((S + 127) // 128)
        parts = [None, None, None]
        if rel is not None:
            parts = [torch.empty(H, nparts, t.shape[1], dtype=torch.float32, device=dev) for t in (rel2d, rel1d, relx)]
        # per-row terms of d c_attn = sum_j P dP from the dQ kernel (no division of delta by the gain: exact at c_attn = 0)
        dgr = torch.empty(B, H, T, dtype=torch.float32, device=dev)
        hip.attn_bwd(q, k, v, pos_q, pos_k, out, dout.contiguous(), lse, delta, dq, dk, dv, dpq, dpk, B, H, T, S, rel=rel,
                     causal=causal, P=P, gain=gain, drel2d_part=parts[0], drel1d_part=parts[1], drelx_part=parts[2],
                     nparts=nparts, dgain_rows=dgr)
        e = torch.empty(0, dtype=torch.float32, device=dev)
        tabs = [p.sum(1) if p is not None else e for p in parts]
        dgain = dgr.sum((0, 2))
        return dq, dk, dv, dpq.float().sum(0).to(pos_q.dtype), dpk.float().sum(0).to(pos_k.dtype), dgain, tabs[0], tabs[1], tabs[2]
    finally:
        hip.set_stream(prev)


@bias_attention_bwd.register_fake
def _(dout, q, k, v, pos_q, pos_k, gain, out, lse, gcode, rel2d, rel1d, relx, P, code_bias, grid_w, causal):
    e = q.new_empty(0, dtype=torch.float32)
    f = lambda t: e if t is None else t.new_empty(t.shape)
    return (q.new_empty(q.shape), k.new_empty(k.shape), v.new_empty(v.shape), pos_q.new_empty(pos_q.shape),
            pos_k.new_empty(pos_k.shape), gain.new_empty(gain.shape, dtype=torch.float32), f(rel2d), f(rel1d), f(relx))


def _attn_setup(ctx, inputs, output):
    q, k, v, pos_q, pos_k, gain, gcode, rel2d, rel1d, relx, P, code_bias, grid_w, causal = inputs
    ctx.save_for_backward(q, k, v, pos_q, pos_k, gain, output[0], output[1], gcode, rel2d, rel1d, relx)
    ctx.meta = (P, code_bias, grid_w, causal)


def _attn_backward(ctx, gout, glse):
    q, k, v, pos_q, pos_k, gain, out, lse, gcode, rel2d, rel1d, relx = ctx.saved_tensors
    P, code_bias, grid_w, causal = ctx.meta
    dq, dk, dv, dpq, dpk, dgain, d2, d1, dx = torch.ops.ifseg.bias_attention_bwd(
        gout, q, k, v, pos_q, pos_k, gain, out, lse, gcode, rel2d, rel1d, relx, P, code_bias, grid_w, causal)
    has = gcode is not None
    return (dq, dk, dv, dpq, dpk, dgain, None, d2 if has else None, d1 if has else None, dx if has else None,
            None, None, None, None)


bias_attention.register_autograd(_attn_backward, setup_context=_attn_setup)
