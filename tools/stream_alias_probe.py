"""Which torch streams share a hardware queue with the default stream (and with each other)?  A long kernel runs on stream A;
a tiny kernel is launched on stream B right after: if B's kernel completes only when A's does, A and B are serialised
(same HW queue).  Prints an aliasing map for the first N pool streams, before and after an RCCL communicator exists."""
import os, sys, time
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1] if len(sys.argv) > 1 else "29661", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
tiny = torch.zeros(64, device=dev)
def long_kernel():
    for _ in range(6):
        torch.mm(big, big)
def aliased(a, b):
    """True if a tiny kernel on b waits for the long work on a"""
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record(); long_kernel(); e1.record()
    with torch.cuda.stream(b):
        tiny.add_(1); e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e2) > 0.5 * e0.elapsed_time(e1)
def survey(tag, streams):
    main = torch.cuda.default_stream()
    row = "".join("X" if aliased(main, s) else "." for s in streams)
    print("%-34s vs default: %s" % (tag, row))
    for i in range(min(6, len(streams))):
        print("   stream %2d vs others: %s" % (i, "".join("-" if j == i else ("X" if aliased(streams[i], streams[j]) else ".") for j in range(len(streams)))))
N = int(os.environ.get("PROBE_N", "12"))
norm = [torch.cuda.Stream() for _ in range(N)]
hi = [torch.cuda.Stream(priority=-1) for _ in range(4)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
survey("normal-priority pool streams", norm)
survey("high-priority pool streams", hi)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
survey("same streams, after an all_reduce", norm)
later = [torch.cuda.Stream() for _ in range(N)]
survey("streams created after it", later)
dist.destroy_process_group()
