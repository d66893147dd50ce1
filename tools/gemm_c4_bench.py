"""Stand-alone GEMM times at SegOFA-Large's shapes (C4: M = 8 x 1839 = 14712 rows, C = 1024, F = 4096): the forward (NT), the dX
(NN) and the dX with the GELU + LayerNorm-backward epilogue, per product -- why the step's NN family costs more than its NT family."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)

def bench(name, fn, flops, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-44s %8.1f us  %7.1f TF/s" % (name, us, flops / us / 1e6), flush=True)

M = int(sys.argv[1]) if len(sys.argv) > 1 else 14712
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    x, w, b = r(M, K), r(N, K), r(N)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bench("NT  y[M,%d] = x[M,%d] w^T" % (N, K), lambda: hip.linear_fwd(x, w, b, out=y), 2.0 * M * N * K)
    dy = r(M, N); dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    bench("NN  dx[M,%d] = dy[M,%d] w" % (K, N), lambda: hip.linear_dx(dy, w, out=dx), 2.0 * M * N * K)
# the fc2 dX with the ffn_layernorm + GELU backward in its epilogue: dy [M, 1024] . W2 [1024, 4096] -> du [M, 4096]
C, F = 1024, 4096
dy, w2, u, du = r(M, C), r(C, F), r(M, F), torch.empty(M, F, dtype=torch.bfloat16, device=dev)
gam, mu, rs, cst = torch.ones(F, device=dev), torch.zeros(M, device=dev), torch.ones(M, device=dev), torch.zeros(M, 2, device=dev)
bench("NN+GLN  du[M,4096] = LN'GELU'(dy[M,1024] W2)", lambda: hip.linear_dx_gelu_ln_bwd(dy, w2, du, u, gam, mu, rs, cst), 2.0 * M * C * F)
bench("NN      dz[M,4096] = dy[M,1024] W2 (plain)", lambda: hip.linear_dx(dy, w2, out=du), 2.0 * M * C * F)
