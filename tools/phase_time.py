"""GPU time of the phases of one training step on the main stream (HIP events, no profiler)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd.tasks.mm_tasks.segmentation import SegmentationTask
from ifseg_amd.criterions.seg_criterion import SegCriterion
from ifseg_amd.trainer import Trainer

dev = torch.device("cuda:0")
task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
model = task.build_model()
model.cfg.dropout, model.cfg.encoder_drop_path_rate, model.cfg.decoder_drop_path_rate = 0.1, 0.1, 0.1
trainer = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
ring = []
for j in range(2):
    sm = task.synthetic_sample(8, dev, seed=1234 + 7919 * j)
    sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
    ring.append(sm)
eng = trainer.eng
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
_f, _b, _j = eng._forward, eng._backward, eng._join_side
def fwd(*a, **k):
    mark("fwd_begin"); r = _f(*a, **k); mark("fwd_end"); return r
def bwd(*a, **k):
    mark("bwd_begin"); r = _b(*a, **k); mark("bwd_end"); return r
jn = [0]
def join():
    mark("join%d_before" % jn[0]); _j(); mark("join%d_after" % jn[0]); jn[0] += 1
eng._forward, eng._backward, eng._join_side = fwd, bwd, join
N = 8
for i in range(3):
    trainer.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
torch.cuda.synchronize()
acc = {}
for i in range(N):
    marks.clear(); jn[0] = 0
    mark("step_begin")
    trainer.train_step([ring[(i + 1) % 2]], prefetch=[ring[i % 2]])
    mark("step_end")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[(n0, n1)] = acc.get((n0, n1), 0.0) + e0.elapsed_time(e1)
tot = 0
for k, v in acc.items():
    print("%-16s -> %-16s %7.3f ms" % (k[0], k[1], v / N)); tot += v / N
print("sum %.3f ms" % tot)
