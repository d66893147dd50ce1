"""Is the eager training step host-bound?  Host time of each (asynchronous) train_step call next to the device time per step.
usage: python tools/host_time.py [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd.criterions import SegCriterion
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
model = task.build_model()
tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev, lazy_logs=True)
ring = []
for j in range(4):
    sm = task.synthetic_sample(8, dev, seed=100 + j)
    sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
    ring.append(sm)
for i in range(10):
    tr.train_step([ring[i % 4]], prefetch=[ring[(i + 1) % 4]])
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(steps):
    a = time.perf_counter()
    tr.train_step([ring[i % 4]], prefetch=[ring[(i + 1) % 4]])
    host.append(time.perf_counter() - a)
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print("steps %d: device %.2f ms/step; host launch loop %.2f ms/step (median call %.2f, min %.2f, max %.2f); host finished %.1f ms before the device"
      % (steps, t_all / steps * 1e3, t_launch / steps * 1e3, host[len(host) // 2] * 1e3, host[0] * 1e3, host[-1] * 1e3, (t_all - t_launch) * 1e3))
