import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from ifseg_amd import hip
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvictim.so"))
mode = sys.argv[1]
s2 = torch.cuda.Stream()
M_, K_, N_ = 4096, 768, 768
x = torch.randn(M_, K_, device=dev).to(torch.bfloat16)
w = torch.randn(N_, K_, device=dev).to(torch.bfloat16)
o = torch.empty(M_, N_, device=dev, dtype=torch.bfloat16)
wt = torch.randn(N_, N_, device=dev).to(torch.bfloat16)
dx = torch.empty(M_, N_, device=dev, dtype=torch.bfloat16)
log = torch.zeros(4096 * 4, dtype=torch.int32, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
kind = sys.argv[2] if len(sys.argv) > 2 else "regs"
src = (torch.arange(3072, device=dev, dtype=torch.int32) + 0x40000000)
srcs = [(torch.arange(3072, device=dev, dtype=torch.int32) + 0x40000000 + e * 0x100000) for e in range(4)]
pw = torch.arange(768 * 1536, device=dev, dtype=torch.int32) + 0x10000000
pg = torch.arange(3072, device=dev, dtype=torch.int32) + 0x40000000
pb = torch.arange(3072, device=dev, dtype=torch.int32) + 0x50000000
src2 = (torch.arange(3072, device=dev, dtype=torch.int32) + 0x40100000)
for it in range(30):
    ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev)
    with torch.cuda.stream(s2):
        prev = hip.set_stream(s2.cuda_stream)
        for _ in range(40):
            if mode == "linear": hip.linear_fwd(x, w, None, out=o)
            elif mode == "dx": hip.linear_dx(o, wt, out=dx)
            elif mode == "torch": torch.mm(x, w.t(), out=o)
        hip.set_stream(prev)
    for rep in range(10):
        if kind == "order":
            lib.victim_order_launch(*[ctypes.c_void_p(t.data_ptr()) for t in srcs], 3072, ctypes.c_void_p(log.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 2304, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        elif kind == "pk":
            lib.victim_pk_launch(ctypes.c_void_p(log.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 2304, 200, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        elif kind == "loads":
            lib.victim_load_launch(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(src2.data_ptr()), 3072, ctypes.c_uint(0x40000000), ctypes.c_void_p(log.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 192 * 12, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        else: lib.victim_launch(ctypes.c_void_p(log.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 512, 40, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
n = int(cnt.item())
print(mode, "corrupted register instances:", n)
lg = log.view(-1, 4)[:min(n, 4096)].cpu()
print(" by array slot:", sorted(collections.Counter(lg[:, 0].tolist()).items()))
print(" lanes (tid%64) histogram:", sorted(collections.Counter((lg[:, 1] % 64).tolist()).items())[:70])
for r in lg[:12].tolist(): print("  slot %d tid %d got %08x want %08x" % (r[0], r[1], r[2] & 0xffffffff, r[3] & 0xffffffff))
