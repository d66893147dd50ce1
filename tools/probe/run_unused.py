import os, sys, ctypes, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvictim.so"))
dev = torch.device("cuda:0")
pats = [0x00000000, 0x3f800000, 0x7fc00000, 0x7f800001, 0x7f800000, 0xff800000, 0x00000001, 0x807fffff, 0xffffffff, 0x3c003c00, 0x7f7fffff, 0xdeadbeef]
p = torch.tensor(pats, dtype=torch.int64).to(torch.int32).to(dev) if False else torch.tensor([x - (1 << 32) if x >= (1 << 31) else x for x in pats], dtype=torch.int32, device=dev)
out = torch.zeros(2 * len(pats), dtype=torch.int32, device=dev)
lib.pk_unused_half_launch(ctypes.c_void_p(p.data_ptr()), len(pats), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().view(-1, 2).tolist()
for x, r in zip(pats, o): print("unused half %08x -> mismatching lanes lo %d hi %d" % (x, r[0], r[1]))
