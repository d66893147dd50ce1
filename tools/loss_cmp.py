"""Fused upsample + cross-entropy kernel on three class counts: time per call, and the outputs saved to argv[1] so that two builds
(IFSEG_LIB=<other libifseg_hip.so>) can be compared bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ifseg_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
out = {}
for nseg, hp in ((15, 32), (150, 32), (171, 40)):
    B, P = 2, hp * hp
    ldl = (nseg + 7) // 8 * 8
    logits = torch.zeros(B, P + 1, ldl, dtype=torch.bfloat16, device=dev)
    logits[:, :, :nseg] = (torch.randn(B, P + 1, nseg, device=dev) * 2).to(torch.bfloat16)
    H = W = hp * 16
    target = torch.randint(0, nseg + 1, (B, H * W), device=dev) + 100
    tp = torch.zeros(B * P * 9 * nseg, device=dev); sp = torch.zeros(B * P * (2 + 3 * nseg), device=dev)
    stats = torch.zeros(2 + 3 * nseg, device=dev); dl = torch.zeros_like(logits); loss = torch.zeros(1, device=dev)
    hip.seg_loss(logits, target, hp, hp, H, W, nseg, 100, tp, sp, stats, dl, loss)
    torch.cuda.synchronize()
    out[nseg] = (loss.cpu(), dl.cpu(), stats.cpu())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): hip.seg_loss(logits, target, hp, hp, H, W, nseg, 100, tp, sp, stats, dl, loss)
    e1.record(); torch.cuda.synchronize()
    print(nseg, "classes: %.1f us per call (tiles + gather), loss %.6f" % (e0.elapsed_time(e1) * 200, loss.item()))
torch.save(out, sys.argv[1])
