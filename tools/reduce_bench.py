"""Stand-alone time of ifseg_attn_bwd_reduce on the encoder shape of SegOFA-Base (B=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd import hip
dev = torch.device("cuda:0")
B, H, T, C = 8, 12, 1060, 768
S = T
nparts = B * ((S + 127) // 128)
n2d, n1d = 63 * 63, 71
r = lambda *s: torch.randn(*s, device=dev)
dpq, dpk = r(B, T, C).bfloat16(), r(B, S, C).bfloat16()
aq, ak = r(T, C), r(S, C)
delta, gain, dgain = r(B, H, T), torch.rand(H, device=dev) + 0.5, torch.zeros(H, dtype=torch.bfloat16, device=dev)
tabs = [(r(H, nparts, n2d), torch.randint(0, 6892, (n2d,), device=dev).int(), r(6892, H)),
        (r(H, nparts, n1d), torch.randint(0, 511, (n1d,), device=dev).int(), r(511, H))]
def run(): hip.attn_bwd_reduce(B, H, T, S, C, dpq, dpk, aq, ak, True, delta, gain, dgain, nparts, tabs)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("attn_bwd_reduce: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
