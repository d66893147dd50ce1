"""Pieces of the ffn_layernorm backward at the Base shapes (M = 8480, J = 768, N = 3072): plain dX GEMM + wide LayerNorm
backward kernel against row statistics + dX GEMM with the GELU/LayerNorm-backward epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip
dev = torch.device("cuda:0")
M, J, N = 8480, 768, 3072
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc)
u = r(M, N).to(torch.bfloat16); gamma = (1 + 0.2 * r(N)).contiguous(); beta = (0.1 * r(N)).contiguous()
w2 = r(J, N, sc=0.05).to(torch.bfloat16); b2 = r(J, sc=0.1).to(torch.bfloat16); dy = r(M, J, sc=0.1).to(torch.bfloat16)
z = torch.empty(M, N, dtype=torch.bfloat16, device=dev); mu, rs = torch.empty(M, device=dev), torch.empty(M, device=dev)
hip.ln_fwd(u, gamma, beta, z, mu, rs, gelu=True)
t = hip.linear_fwd(z, w2, b2)
coef = torch.empty(2, J, device=dev); c = torch.empty(M, 2, device=dev)
du = torch.empty(M, N, dtype=torch.bfloat16, device=dev); dz = torch.empty_like(du)
part = torch.empty(2, hip.LN_BWD_BLOCKS, N, device=dev)
buf = torch.empty(J * N + J, dtype=torch.bfloat16, device=dev); dw2, db2 = buf[: J * N].view(J, N), buf[J * N:]
dg, db = torch.empty(N, dtype=torch.bfloat16, device=dev), torch.empty(N, dtype=torch.bfloat16, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("plain dX GEMM (dz)            %.1f us" % timeit(lambda: hip.linear_dx(dy, w2, out=dz)))
print("wide ln_bwd (gelu)            %.1f us" % timeit(lambda: hip.ln_bwd(dz, u, gamma, mu, rs, du, part[0], part[1], gelu=True)))
print("ffn_ln_coef                   %.1f us" % timeit(lambda: hip.ffn_ln_coef(w2, gamma, beta, b2, coef)))
print("ffn_ln_rowstats               %.1f us" % timeit(lambda: hip.ffn_ln_rowstats(dy, t, coef, c, N)))
print("dX GEMM + GELU/LN epilogue    %.1f us" % timeit(lambda: hip.linear_dx_gelu_ln_bwd(dy, w2, du, u, gamma, mu, rs, c)))
print("ffn_ln_param_grads            %.1f us" % timeit(lambda: hip.ffn_ln_param_grads(w2, dw2, db2, gamma, beta, dg, db)))
