"""GEMM micro-benchmark on the SegOFA-Base shapes (B=8): TF/s per layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

def bench(name, fn, flops, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-34s %8.1f us  %7.1f TF/s" % (name, us, flops / us / 1e6))

def main():
    dev = torch.device("cuda:0")
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    M = 8480
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        x, w, b = r(M, K), r(N, K), r(N)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bench("NT  M%d N%d K%d" % (M, N, K), lambda: hip.linear_fwd(x, w, b, out=y), 2.0 * M * N * K)
        dy = r(M, N); dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
        bench("NN  M%d N%d K%d" % (M, K, N), lambda: hip.linear_dx(dy, w, out=dx), 2.0 * M * N * K)
        dw = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
        bench("TN  M%d N%d K%d" % (N, K, M), lambda: hip.linear_dw(dy, x, dw), 2.0 * M * N * K)
    # reference point: a square problem
    for n in (4096,):
        a, b2 = r(n, n), r(n, n); c = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
        bench("NT  %d^3" % n, lambda: hip.linear_fwd(a, b2, out=c), 2.0 * n ** 3)
        bench("torch.matmul %d^3 (hipBLASLt)" % n, lambda: torch.matmul(a, b2.t(), out=c), 2.0 * n ** 3)
    x, w = r(M, 768), r(3072, 768); y = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
    bench("torch F.linear M8480 N3072 K768", lambda: torch.matmul(x, w.t(), out=y), 2.0 * M * 3072 * 768)
if __name__ == "__main__":
    main()
