"""Stand-alone time of the fused clip + Adam launch over an arena of SegOFA-Base's size (109 M parameters: fp32 master, m, v read and
written, bf16 gradient read, bf16 copy written = 28 bytes per parameter) next to a torch copy of the same number of bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip

def main():
    dev = torch.device("cuda:0")
    n = 109_300_000
    p32 = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    g = (torch.randn(n, device=dev) * 1e-3).to(torch.bfloat16); p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ss = torch.ones(1, device=dev)
    def step(k):
        hip.adam_step(p32, g, m, v, p16, 1e-4, 0.9, 0.999, 1e-8, 0.01, k, 1.0, 1.0, ss)
    for k in range(1, 4): step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(4, 24): step(k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-28s adam %7.1f us  %6.0f GB/s" % (os.environ.get("IFSEG_LIB", "default").split("/")[-1], us, n * 28 / us / 1e3))
    a = torch.empty(n * 14 // 4, device=dev); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print("%-28s copy %7.1f us  %6.0f GB/s (same bytes, read + write)" % ("", us, n * 28 / us / 1e3))

if __name__ == "__main__":
    main()
