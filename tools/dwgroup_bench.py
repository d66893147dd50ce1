"""Stand-alone time of the grouped weight-gradient GEMM of one encoder layer (q|k|v, out_proj, fc1, fc2), by grid cap."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
M, C, F = 8480, 768, 3072
tasks = []
for (n, k) in ((3 * C, C), (C, C), (F, C), (C, F)):
    dy, x = r(M, n), r(M, k)
    buf = torch.empty(n * k + n, dtype=torch.bfloat16, device=dev)
    tasks.append((dy, x, buf[:n * k].view(n, k), buf[n * k:]))
flops = sum(2.0 * M * t[0].shape[1] * t[1].shape[1] for t in tasks)
for cap in (256, 384, 512, 768, 1024):
    hip.DW_GROUP_WGS = cap
    hip.linear_dw_group(tasks); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.linear_dw_group(tasks)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print("cap %4d: %7.1f us  %6.1f TF/s" % (cap, us, flops / us / 1e6))

