"""Micro-benchmark of the LayerNorm kernels on the SegOFA-Base row shapes (B=8): effective GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip


def bench(name, fn, nbytes, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-40s %8.1f us  %7.1f GB/s" % (name, us, nbytes / us / 1e3))


def main():
    dev = torch.device("cuda:0")
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    M = 8480
    for C, gelu, add in ((768, False, True), (3072, True, False)):
        x, dy, res = r(M, C), r(M, C), r(M, C)
        g, b = r(C), r(C)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        part = torch.empty(2, hip.LN_BWD_BLOCKS, C, device=dev)
        bench("ln_fwd C%d gelu%d resid%d" % (C, gelu, add),
              lambda: hip.ln_fwd(x, g, b, y, mean, rstd, resid=res if add else None, gelu=gelu), M * C * (6.0 if add else 4.0))
        bench("ln_bwd C%d gelu%d add%d" % (C, gelu, add),
              lambda: hip.ln_bwd(dy, x, g, mean, rstd, dx, part[0], part[1], dx_add=res if add else None, gelu=gelu),
              M * C * (8.0 if add else 6.0))
        if C <= 1024:
            y2 = torch.empty_like(x); m2, r2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
            gf, bf = g.float(), b.float()
            bench("ln_fwd_pair C%d resid (fp32 params)" % C,
                  lambda: hip.ln_fwd_pair(x, gf, bf, y, mean, rstd, gf, bf, y2, m2, r2, resid=res), M * C * 8.0)
            dx2 = torch.empty_like(x)
            bench("ln_bwd_drop C%d add (fp32 params)" % C,
                  lambda: hip.ln_bwd_drop(dy, x, gf, mean, rstd, dx, part[0], part[1], dx2, dx_add=res), M * C * 10.0)
        gw = torch.empty(2, C, device=dev)
        bench("reduce_parts 2x%dx%d" % (hip.LN_BWD_BLOCKS, C), lambda: hip.reduce_parts(part, gw, 2, hip.LN_BWD_BLOCKS, C), 2 * hip.LN_BWD_BLOCKS * C * 4.0)
        cs = torch.empty(hip.COLSUM_BLOCKS, C, device=dev)
        bench("colsum M%d C%d" % (M, C), lambda: hip.colsum(x, cs), M * C * 2.0)
    n = 106_000_000
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b2 = torch.empty_like(a)
    bench("torch copy bf16 %dM (HBM reference)" % (n // 1_000_000), lambda: b2.copy_(a), n * 4.0, iters=10)


if __name__ == "__main__":
    main()
