"""Micro-benchmark of the batch-inner attention backward (csrc/attention_bi.hip) next to the round-3 kernels, on the
encoder / decoder / cross shapes of SegOFA-Base (B=8) or SegOFA-Large (ATTN_BENCH_LARGE=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main(kind="enc"):
    dev = torch.device("cuda:0")
    B, H, C = 8, 12, 768
    gh = gw = 32; P = 1024
    Lt = 1 if kind in ("dec", "decfull") else 36
    if os.environ.get("ATTN_BENCH_LARGE"):
        B, H, C = 4, 16, 1024
        gh = gw = 40; P = 1600
        Lt = 1 if kind in ("dec", "decfull") else 239
    if os.environ.get("ATTN_BENCH_B"):
        B = int(os.environ["ATTN_BENCH_B"])
    if os.environ.get("ATTN_BENCH_LT"):
        Lt = int(os.environ["ATTN_BENCH_LT"])
    T = S = P + Lt
    causal = kind == "dec"
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    qkv = r(B, T, 3 * C); pq, pk = r(T, C), r(S, C); dout = r(B, T, C)
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    gcode = (ys * (2 * gw - 1) + xs).reshape(-1).int().to(dev)
    n2d = (2 * gh - 1) * (2 * gw - 1)
    rel = hip.RelBias(P, gcode, (gh - 1) * (2 * gw - 1) + gw - 1, torch.randn(H, n2d, generator=g).to(dev),
                      torch.randn(H, 2 * Lt - 1, generator=g).to(dev), torch.randn(H, 2, generator=g).to(dev), grid_w=gw)
    if kind == "cross":
        rel = None
    gain = torch.ones(H, device=dev)
    out = torch.zeros(B, T, C, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, T, device=dev)
    dqkv = torch.zeros_like(qkv); delta = torch.zeros(B, H, T, device=dev)
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    dq, dk, dv = dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:]
    hip.attn_fwd(q, k, v, pq, pk, out, lse, B, H, T, S, rel=rel, causal=causal, gain=gain)
    delta.copy_((dout.float() * out.float()).view(B, T, H, 64).sum(-1).permute(0, 2, 1))
    dense = hip.DenseBias(H, T, S, dev)
    ng = (B + 3) // 4
    dbias = torch.zeros(ng, H, T, dense.Sp, dtype=torch.bfloat16, device=dev)
    dpq = torch.zeros(T, C, device=dev); dpk = torch.zeros(S, C, device=dev)
    kw = {}
    if rel is not None:
        NP = hip.dbias_nparts()
        kw = dict(P=P, grid_h=gh, grid_w=gw, drel2d=torch.zeros(H, NP, n2d, device=dev),
                  drel1d=torch.zeros(H, NP, 2 * Lt - 1, device=dev), drelx=torch.zeros(H, NP, 2, device=dev))
    bi = lambda ph: hip.attn_bwd_bi(q, k, v, dout, lse, delta, dense, dq, dk, dv, dbias, B, H, T, S, causal=causal,
                                    P=P if (rel is not None or causal) else None, gain=gain, phases=ph)
    out2, lse2 = torch.zeros_like(out), torch.zeros_like(lse)
    print(kind, "round-3 fwd             us %.1f" % timeit(lambda: hip.attn_fwd(q, k, v, pq, pk, out, lse, B, H, T, S, rel=rel, causal=causal, gain=gain)))
    print(kind, "dense bias build        us %.1f" % timeit(lambda: hip.attn_dense_bias(dense, pq, pk, rel=rel, causal=causal, P=P)))
    print(kind, "bi fwd                  us %.1f" % timeit(lambda: hip.attn_fwd_bi(q, k, v, dense, out2, lse2, B, H, T, S, causal=causal, P=P if (rel is not None or causal) else None, gain=gain)))
    print(kind, "bi dkv                  us %.1f" % timeit(lambda: bi(hip.ATTN_BWD_DKV)))
    print(kind, "bi dq (+ sum_b dS)      us %.1f" % timeit(lambda: bi(hip.ATTN_BWD_DQ)))
    print(kind, "bi dkv + dq             us %.1f" % timeit(lambda: bi(0)))
    print(kind, "dbias grads             us %.1f" % timeit(lambda: hip.attn_dbias_grads(dbias, S, pos_q=pq, pos_k=pk, dpq_acc=dpq, dpk_acc=dpk, causal=causal, **kw)))
    print(kind, "  operands only         us %.1f" % timeit(lambda: hip.attn_dbias_grads(dbias, S, pos_q=pq, pos_k=pk, dpq_acc=dpq, dpk_acc=dpk, causal=causal, P=P)))
    if kw:
        print(kind, "  delta tables only     us %.1f" % timeit(lambda: hip.attn_dbias_grads(dbias, S, causal=causal, **kw)))
    # round-3 kernels
    dpqp = torch.zeros(B, T, C, device=dev, dtype=torch.bfloat16); dpkp = torch.zeros(B, S, C, device=dev, dtype=torch.bfloat16)
    nparts = B * ((S + 127) // 128)
    parts = [torch.zeros(H, nparts, n, device=dev) for n in (n2d, 2 * Lt - 1, 2)] if rel is not None else [None] * 3
    old = lambda ph: hip.attn_bwd(q, k, v, pq, pk, out, dout, lse, delta, dq, dk, dv, dpqp, dpkp, B, H, T, S, rel=rel,
                                  causal=causal, gain=gain, drel2d_part=parts[0], drel1d_part=parts[1],
                                  drelx_part=parts[2], nparts=nparts, phases=ph)
    print(kind, "round-3 dkv             us %.1f" % timeit(lambda: old(hip.ATTN_BWD_DKV)))
    print(kind, "round-3 dq              us %.1f" % timeit(lambda: old(hip.ATTN_BWD_DQ)))
    print(kind, "round-3 dkv + dq        us %.1f" % timeit(lambda: old(hip.ATTN_BWD_DKV | hip.ATTN_BWD_DQ)))


if __name__ == "__main__":
    for kd in (sys.argv[1:] or ["enc", "dec", "cross"]):
        main(kd)
