"""Micro-benchmark of the implicit-GEMM convolution on the ResNet-101 trunk shapes (B=8, 512x512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip


def main():
    dev = torch.device("cuda:0")
    r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
    B = int(os.environ.get("CONV_BENCH_B", "8"))
    shapes = [  # (H, Cin, Cout, k, stride, count per step)
        (128, 64, 64, 1, 1, 3), (128, 64, 64, 3, 1, 3), (128, 64, 256, 1, 1, 4), (128, 256, 64, 1, 1, 2),
        (128, 256, 128, 1, 1, 1), (128, 128, 128, 3, 2, 1), (64, 128, 512, 1, 1, 4), (64, 512, 128, 1, 1, 3),
        (64, 128, 128, 3, 1, 3), (128, 256, 512, 1, 2, 1),
        (64, 512, 256, 1, 1, 1), (64, 256, 256, 3, 2, 1), (32, 256, 1024, 1, 1, 23), (32, 1024, 256, 1, 1, 22),
        (32, 256, 256, 3, 1, 22), (64, 512, 1024, 1, 2, 1),
    ]
    total = 0.0
    for (H, ci, co, k, st, cnt) in shapes:
        x = r(B, H, H, ci); w = r(co, k, k, ci); sh = r(co)
        Ho = (H + 2 * (k // 2) - k) // st + 1
        out = torch.empty(B, Ho, Ho, co, dtype=torch.bfloat16, device=dev)
        fn = lambda: hip.conv2d_nhwc(x, w, sh, None, out, B, H, H, ci, co, k, k, st, k // 2, True)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        fl = 2.0 * B * Ho * Ho * co * k * k * ci
        total += us * cnt
        print("H%3d %4d->%4d k%d s%d  x%2d  %7.1f us  %6.1f TF/s" % (H, ci, co, k, st, cnt, us, fl / us / 1e6))
    print("trunk convs total %.2f ms per step" % (total / 1e3))


if __name__ == "__main__":
    main()
