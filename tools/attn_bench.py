"""Micro-benchmark of the attention kernels on the encoder / decoder shapes of SegOFA-Base (B=8)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

def main(kind="enc", iters=5):
    dev = torch.device("cuda:0")
    B, H, C = 8, 12, 768
    gh = gw = 32; P = 1024
    Lt = 1 if kind in ("dec", "decfull") else 36
    if os.environ.get("ATTN_BENCH_LARGE"):      # BASELINE configs[3] geometry: 640 x 640 -> 40 x 40 grid, 16 heads, L = 239
        B, H, C = 4, 16, 1024
        gh = gw = 40; P = 1600
        Lt = 1 if kind in ("dec", "decfull") else 239
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
    gain = torch.ones(H, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(B, T, C, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, T, device=dev)
    dqkv = torch.zeros_like(qkv); delta = torch.zeros(B, H, T, device=dev)
    dpq = torch.zeros(B, T, C, device=dev, dtype=torch.bfloat16); dpk = torch.zeros(B, S, C, device=dev, dtype=torch.bfloat16)
    nparts = B * ((S + 127) // 128)
    parts = [torch.zeros(H, nparts, n, device=dev) for n in (n2d, 2 * Lt - 1, 2)] if rel is not None else [None] * 3
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    def step():
        hip.attn_fwd(q, k, v, pq, pk, out, lse, B, H, T, S, rel=rel, causal=causal, gain=gain)
        hip.attn_bwd(q, k, v, pq, pk, out, dout, lse, delta, dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:], dpq, dpk,
                     B, H, T, S, rel=rel, causal=causal, gain=gain, drel2d_part=parts[0], drel1d_part=parts[1],
                     drelx_part=parts[2], nparts=nparts)
    step(); torch.cuda.synchronize()
    hip.prof_reset(); hip.prof_enable(0x70)
    t0 = time.time()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    for kd in (4, 5, 6):
        p = hip.prof_read(kd)
        print(kind, p["kind"], "avg us %.1f" % (p["ms"] * 1e3 / max(1, p["launches"])), "alg TF/s %.1f" % (p["flops"] / max(1e-9, p["ms"] * 1e-3) / 1e12))
    hip.prof_enable(0)
    def bwd(phases):
        hip.attn_bwd(q, k, v, pq, pk, out, dout, lse, delta, dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:], dpq, dpk,
                     B, H, T, S, rel=rel, causal=causal, gain=gain, drel2d_part=parts[0], drel1d_part=parts[1],
                     drelx_part=parts[2], nparts=nparts, phases=phases)
    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    print(kind, "dkv alone            us %.1f" % timeit(lambda: bwd(hip.ATTN_BWD_DKV)))
    print(kind, "dq kernel alone      us %.1f" % timeit(lambda: bwd(hip.ATTN_BWD_DQ)))
if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "enc")
