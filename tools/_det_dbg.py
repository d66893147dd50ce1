import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ifseg_amd.criterions import SegCriterion
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
dev = torch.device("cuda:0")
def run(overlap=True, steps=1):
    torch.manual_seed(0)
    task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
    model = task.build_model()
    tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
    tr.eng.overlap = overlap
    ring = []
    for j in range(2):
        sm = task.synthetic_sample(2, dev, seed=100 + j)
        sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
        ring.append(sm)
    for i in range(steps):
        logs = tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]] if overlap else None)
    torch.cuda.synchronize()
    eng = tr.eng
    return float(logs[-1]["loss"]), {n: eng.G(n).clone() for n in eng.trainable_names()}
for steps in (2,):
    l1, g1 = run(True, steps); l2, g2 = run(True, steps); l3, g3 = run(False, steps)
    print("steps", steps, "loss", l1, l2, l3)
    for tag, gb in (("overlap#2", g2), ("no-overlap", g3)):
        bad = [n for n in g1 if not torch.equal(g1[n], gb[n])]
        print("  vs", tag, len(bad), "tensors differ:", bad[:8])
