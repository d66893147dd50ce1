"""Host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
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
sample = task.synthetic_sample(8, dev, seed=1234)
sample["net_input"]["patch_images"] = sample["net_input"]["patch_images"].to(torch.bfloat16)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_step([sample])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step   wall %.2f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        trainer.train_step([sample])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# host seconds spent inside the engine's forward / backward python (enqueue only)
eng = trainer.eng
acc = {"fwd": 0.0, "bwd": 0.0}
_f, _b = eng.forward, eng.backward
def fwd(*a, **k):
    t = time.perf_counter(); r = _f(*a, **k); acc["fwd"] += time.perf_counter() - t; return r
def bwd(*a, **k):
    t = time.perf_counter(); r = _b(*a, **k); acc["bwd"] += time.perf_counter() - t; return r
eng.forward, eng.backward = fwd, bwd
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_step([sample])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host: forward %.2f ms  backward %.2f ms  rest %.2f ms  (per step)" % (acc["fwd"] / N * 1e3, acc["bwd"] / N * 1e3, ((t1 - t0) - acc["fwd"] - acc["bwd"]) / N * 1e3))
# same with the GPU drained before each phase is timed: pure python/launch cost when the queue is empty
acc = {"fwd": 0.0, "bwd": 0.0}
def fwd2(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _f(*a, **k); acc["fwd"] += time.perf_counter() - t; return r
def bwd2(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _b(*a, **k); acc["bwd"] += time.perf_counter() - t; return r
eng.forward, eng.backward = fwd2, bwd2
for _ in range(N):
    trainer.train_step([sample])
torch.cuda.synchronize()
print("host (queue drained first): forward %.2f ms  backward %.2f ms" % (acc["fwd"] / N * 1e3, acc["bwd"] / N * 1e3))
