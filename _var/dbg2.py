import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_kernels_gpu as t
from ifseg_amd import hip
dev = torch.device("cuda:0")
H, B = 3, 2
gh, gw, P, Lt = 32, 32, 1024, 36
T = S = P + Lt
C = H * 64
q, k, v = t._rand((B, T, C), dev, 20, 0.35), t._rand((B, S, C), dev, 21), t._rand((B, S, C), dev, 22)
pq, pk = t._rand((T, C), dev, 23, 0.35), t._rand((S, C), dev, 24)
dout = t._rand((B, T, C), dev, 25)
gain = torch.ones(H, device=dev, dtype=torch.bfloat16)
gcode, code_bias, n2d = t._grid_codes(gh, gw)
g = torch.Generator().manual_seed(30)
tabs = [torch.randn(H, n2d, generator=g), torch.randn(H, 2 * Lt - 1, generator=g), torch.randn(H, 2, generator=g)]
rel = hip.RelBias(P, gcode.to(dev), code_bias, tabs[0].to(dev), tabs[1].to(dev), tabs[2].to(dev), grid_w=gw)
out = torch.zeros(B, T, C, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, T, device=dev)
hip.attn_fwd(q, k, v, pq, pk, out, lse, B, H, T, S, rel=rel, gain=gain)
dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.full_like(v, 7.0)
delta = torch.zeros(B, H, T, device=dev); dpq = torch.zeros(B, T, C, device=dev); dpk = torch.zeros(B, S, C, device=dev)
nparts = B * ((S + 127) // 128)
parts = [torch.zeros(H, nparts, n, device=dev) for n in (n2d, 2 * Lt - 1, 2)]
hip.attn_bwd(q, k, v, pq, pk, out, dout, lse, delta, dq, dk, dv, dpq, dpk, B, H, T, S, rel=rel, gain=gain,
             drel2d_part=parts[0], drel1d_part=parts[1], drelx_part=parts[2], nparts=nparts)
torch.cuda.synchronize()
qh = q.float().view(B, T, H, 64).transpose(1, 2); kh = k.float().view(B, S, H, 64).transpose(1, 2)
sc = qh @ kh.transpose(2, 3) + pq.float().view(T, H, 64).transpose(0, 1) @ pk.float().view(S, H, 64).permute(1, 2, 0)
sc = sc + t._dense_rel(H, T, S, P, gcode.long(), code_bias, *tabs).to(dev)
pr = torch.softmax(sc, -1)                      # [B,H,T,S]
doh = dout.float().view(B, T, H, 64).transpose(1, 2)
vh = v.float().view(B, S, H, 64).transpose(1, 2)
dP = doh @ vh.transpose(2, 3)
dS = pr * (dP - delta.unsqueeze(-1))
dvh = dv.float().view(B, S, H, 64).transpose(1, 2)   # [B,H,S,64]
cands = {"P^T dO": pr.transpose(2, 3) @ doh, "dS^T dO": dS.transpose(2, 3) @ doh, "expS^T dO": sc.exp().transpose(2, 3) @ doh,
         "P^T Q": pr.transpose(2, 3) @ qh, "dP^T dO": dP.transpose(2,3) @ doh}
for n, c in cands.items():
    print(n, "rel err", ((dvh[:, :, :P] - c[:, :, :P]).norm() / c[:, :, :P].norm()).item())
x = dvh[0, 0, :4, :8]; y = cands["P^T dO"][0, 0, :4, :8]
print(x); print(y)
