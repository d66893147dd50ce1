"""Two runs of N training steps (Base, B = 8, every stream on, trunk look-ahead): per-step loss and the final parameters must be
bit-equal.  A long version of tests/test_model_gpu.py::test_training_step_is_deterministic_across_streams -- sporadic
corruption of a kernel under concurrency (see tools/probe/README.md) at a rate the 3-step test would miss shows up here.
usage: python tools/long_determinism.py [steps=40] [batch=8] [base|large]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifseg_amd.criterions import SegCriterion
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
large = len(sys.argv) > 3 and sys.argv[3] == "large"
task = (SegmentationTask(num_seg_tokens=171, patch_image_size=640, arch="segofa_large") if large
        else SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base"))
ring = []
for j in range(4):
    sm = task.synthetic_sample(B, dev, seed=100 + j)
    sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
    ring.append(sm)
def run():
    torch.manual_seed(0)
    tr = Trainer(task.build_model(), SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
    losses = []
    for i in range(steps):
        lg = tr.train_step([ring[i % 4]], prefetch=[ring[(i + k) % 4] for k in range(1, 4)])
        losses.append(lg[0]["loss"].clone() if torch.is_tensor(lg[0]["loss"]) else lg[0]["loss"])
    torch.cuda.synchronize()
    return [float(x) for x in losses], tr.p32.clone(), tr.eng.g16.clone()
l1, p1, g1 = run()
l2, p2, g2 = run()
bad = [i for i in range(steps) if l1[i] != l2[i]]
print("steps %d: losses differ at %s; masters equal %s; last gradient equal %s" % (steps, bad[:10] if bad else "no step", torch.equal(p1, p2), torch.equal(g1, g2)))
sys.exit(0 if not bad and torch.equal(p1, p2) and torch.equal(g1, g2) else 1)
