"""7x7 stem convolution: direct fp32 kernel against the matrix-core kernel, 16 images of 512 x 512 (one trunk pass of two batches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip
dev = torch.device("cuda:0")
B, H, W = 16, 512, 512
x4 = torch.randn(B, H, W, 4, device=dev).to(torch.bfloat16); x4[..., 3] = 0
w = torch.randn(7, 7, 3, 64, device=dev) * 0.1; shift = torch.randn(64, device=dev) * 0.1
out = torch.empty(B, H // 2, W // 2, 64, dtype=torch.bfloat16, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
wt = hip.stem_weights_mfma(w)
print("direct fp32 stem      %.1f us" % timeit(lambda: hip.stem_conv(x4, w, shift, out, B, H, W)))
print("matrix-core stem      %.1f us  (134 MB written: %.0f GB/s)" % ((t := timeit(lambda: hip.stem_conv(x4, wt, shift, out, B, H, W))), 134.2e6 / t / 1e3))
