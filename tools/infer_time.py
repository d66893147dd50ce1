"""Latency of one inference forward (BASELINE config 5: Base, batch 1, 512x512): host enqueue vs wall."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd.tasks.mm_tasks.segmentation import SegmentationTask

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# optional image size "HxW" (default 512x512): anything but 512x512 takes the resized-grid path the reference's validation
# uses at native aspect ratio (criterions/seg_criterion.py:194-217), e.g. 512x683 -> a 32 x 43 feature grid
HW = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (512, 512)
task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
model = task.build_model().to(dev).eval()
sm = task.synthetic_sample(B, dev, seed=1, image_hw=HW)
sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
if HW != (512, 512):
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model(**sm["net_input"]); torch.cuda.synchronize()
        print("%dx%d first forward (builds and caches the dense resized biases): %.1f ms" % (HW[0], HW[1], (time.perf_counter() - t0) * 1e3))
with torch.no_grad():
    for _ in range(5):
        model(**sm["net_input"])
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N):
        model(**sm["net_input"])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("B=%d %dx%d  host enqueue %.2f ms  wall %.2f ms per forward (%.1f img/s)" % (B, HW[0], HW[1], (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, B * N / (t2 - t0)))
if HW != (512, 512):
    sys.exit(0)

# the same forward replayed from a captured graph (static input / output buffers)
g = torch.cuda.CUDAGraph()
static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sm["net_input"].items()}
with torch.no_grad():
    model(**static)                               # warm the per-tensor checks on the static inputs
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out_static = model(**static)[0]
    torch.cuda.synchronize()
    ref = model(**sm["net_input"])[0].clone()
    for k, v in sm["net_input"].items():
        if torch.is_tensor(v):
            static[k].copy_(v)
    g.replay()
    torch.cuda.synchronize()
    print("graph replay equals eager:", torch.equal(out_static, ref))
    t0 = time.perf_counter()
    for _ in range(N):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("B=%d  GRAPH  host %.2f ms  wall %.2f ms per forward (%.1f img/s)" % (B, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, B * N / (t2 - t0)))
