"""ifseg_ln_bwd / ifseg_ln_bwd_drop on fixed inputs: `save` writes the outputs to /tmp, `cmp` compares the current kernels against them
(run once with IFSEG_LN_BWD_CLASSIC=1 save, once without + cmp: the register-lean kernels must reproduce the former ones bit for bit)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ifseg_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
outs = {}
for rows, C in ((8480, 768), (333, 1024), (64, 128), (1001, 256)):
    x, dy, add = [torch.randn(rows, C, device=dev, generator=g).bfloat16() for _ in range(3)]
    gam = torch.randn(C, device=dev, generator=g)
    mu, rs = x.float().mean(1), torch.rsqrt(x.float().var(1, unbiased=False) + 1e-5)
    for ci, (a, dr) in enumerate(((None, None), (add, None), (add, (0.1, 77, None, None)))):
        dx = torch.empty_like(x); part = torch.zeros(2, hip.LN_BWD_BLOCKS, C, device=dev)
        hip.ln_bwd(dy, x, gam, mu, rs, dx, part[0], part[1], dx_add=a, drop=dr)
        dx1 = torch.empty_like(x); dx2 = torch.empty_like(x); part2 = torch.zeros(2, hip.LN_BWD_BLOCKS, C, device=dev)
        hip.ln_bwd_drop(dy, x, gam, mu, rs, dx1, part2[0], part2[1], dx2, dx_add=a, drop2=dr)
        torch.cuda.synchronize()
        for nme, t in (("dx", dx), ("part", part), ("drop.dx", dx1), ("drop.dx2", dx2), ("drop.part", part2)):
            outs["%dx%d case%d %s" % (rows, C, ci, nme)] = t.cpu()
if sys.argv[1] == "save":
    torch.save(outs, "/tmp/ln_ref.pt")
else:
    ref = torch.load("/tmp/ln_ref.pt")
    bad = [(k, (outs[k].float() - ref[k].float()).abs().max().item(), int((outs[k] != ref[k]).sum())) for k in outs if not torch.equal(outs[k], ref[k])]
    print("LN_BWD_CMP", "bit-equal" if not bad else bad[:12])
