import os, sys, time
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mb in (1, 16, 50, 213):
    t = torch.randn(mb * (1 << 20) // 2, device=dev).to(torch.bfloat16)
    for asy in (False, True):
        dist.all_reduce(t); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.time()
        e0.record()
        for _ in range(10):
            if asy:
                w = dist.all_reduce(t, async_op=True); w.wait()
            else:
                dist.all_reduce(t)
        e1.record(); h1 = time.time(); torch.cuda.synchronize()
        print("%4d MB async=%s: %.3f ms GPU per call, %.3f ms host per call" % (mb, asy, e0.elapsed_time(e1) / 10, (h1 - h0) * 100))
dist.destroy_process_group()
