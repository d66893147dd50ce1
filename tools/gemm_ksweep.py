"""Per-tile fixed cost of the GEMM kernels: time of M8480 x N x K for growing K (t = a + b K)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
M = 8480
for N in (768, 2304, 3072):
    row = []
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        x, w, b = r(M, K), r(N, K), r(N)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        us = t(lambda: hip.linear_fwd(x, w, b, out=y))
        row.append("K%d %.1f" % (K, us))
    print("NT N%d:" % N, "  ".join(row))
    row = []
    for K in (64, 128, 256, 512, 768, 1536, 3072):       # NN: dx[M,N] = dy[M,K] . W[K,N]
        dy, w = r(M, K), r(K, N)
        dx = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        us = t(lambda: hip.linear_dx(dy, w, out=dx))
        row.append("K%d %.1f" % (K, us))
    print("NN N%d:" % N, "  ".join(row))
